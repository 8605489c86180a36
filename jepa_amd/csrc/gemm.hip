// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * B[N,K]^T   (both operands K-contiguous, fp32 accumulate)
//
// This one core serves every Linear of the V-JEPA step (reference: nn.Linear call sites modules.py:31-34,63,76;
// predictor.py:194,237; Conv3d patch embed patch_embed.py:56 after tubelet packing):
//   fwd    Y  = X  * W^T            A = X  [M,K],      B = W   [N,K]
//   dgrad  dX = dY * W              A = dY [M,N],      B = W^T [K,N]   (bf16 transposed weight shadow)
//   wgrad  dW = dY^T * X            A = dY^T [N,Mpad], B = X^T [K,Mpad] -> fp32 straight into the grad arena
//
// Structure: 128x128 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles; K-tile 64 (32 as a
// fallback when K % 64 != 0); two LDS buffers; ONE barrier per K-tile; tile t+1 is staged while tile t is
// multiplied.  Staging is LDS-DMA (global_load_lds_dwordx4: no VGPR round trip) with the XOR swizzle applied on the
// per-lane SOURCE address (the DMA destination is lane-linear).  The MFMA is issued with swapped operands (D = Bfrag x Afrag) so each lane owns four
// CONSECUTIVE output columns: 8-byte bf16 / 16-byte fp32 epilogue accesses for C, bias, residual and aux.
// Workgroup ids are remapped so every XCD (private L2) works on a contiguous band of tiles.
#include "gemm_common.hpp"
#include "options.hpp"
#include <cstdlib>

// tile configurations: <BM, BN, WM, WN> = block tile and wave grid; each wave owns (BM/WM) x (BN/WN)
//   128x128, 2x2 waves (64x64 per wave)  : 64 KB LDS, 2 workgroups / CU -- small / skinny problems
//   256x256, 2x4 waves (128x64 per wave) : 128 KB LDS, 1 workgroup / CU, 2 waves / SIMD -- 2x the L2->LDS reuse
//   256x128, 2x2 waves (128x64 per wave) : BK32 x 3 stages = 72 KB, 2 INDEPENDENT workgroups / CU: one workgroup's
//                                          prologue / epilogue overlaps the other's K loop (short-K shapes)

// 16-byte-chunk XOR swizzle of a tile row (rows are BK*2 bytes): makes every ds_read_b128 lane group of a fragment
// read hit 16 distinct 16-byte slots of the 256-byte bank row.  BK=64 (8 chunks/row): chunk ^= row&7.
// BK=32 (4 chunks/row, 4 rows per bank row): chunk ^= (4 - ((row>>2)&3)) & 3.
template <int BK>
__device__ __forceinline__ int swz_of(int row) {
  return BK == 64 ? (row & 7) : ((4 - ((row >> 2) & 3)) & 3);
}

// ---- staging: one 128 x BK bf16 operand tile -> LDS (lane-linear image, source-side swizzle) ----
template <int BK, int ROWS, int NT>
__device__ __forceinline__ void stage_issue(const bf16_t* __restrict__ G, int64_t ld, int64_t row0, int64_t rows,
                                            int64_t k0, char* lds_tile, int tid, int wave_u) {
  constexpr int CPR = BK / 8;                  // 16-byte chunks per tile row
  constexpr int NIT = (ROWS * CPR) / NT;       // chunks per thread
#pragma unroll
  for (int j = 0; j < NIT; j++) {
    const int q = j * NT + tid;
    const int row = q / CPR, cpos = q % CPR;
    const int c = cpos ^ swz_of<BK>(row);
    int64_t gr = row0 + row;
    gr = gr < rows ? gr : rows - 1;            // clamp: tail rows are never stored
    const bf16_t* src = G + gr * ld + k0 + c * 8;
    char* dst = lds_tile + (j * NT + wave_u * 64) * 16;  // wave-uniform base; HW adds lane*16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 0 ? N : 0) : "memory");
}

// NS = LDS ring depth.  NS == 2 is the classic double buffer (stage t+1 while multiplying t, drain before the
// barrier).  NS > 2 keeps NS-2 stages of LDS-DMA in flight ACROSS the per-K-tile barrier with a counted
// s_waitcnt vmcnt (never 0 in steady state) and a raw s_barrier, which is what hides HBM/L2 latency when the
// K loop is short (K = 1024 / 384 in this model).
template <int BK, int NS, int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 ? 2 : 1)) void gemm_nt_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = WM * WN * 64;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int CPR = BK / 8;
  constexpr int NIT_A = (BM * CPR) / NT, NIT_B = (BN * CPR) / NT;
  constexpr int FM = BM / WM / 16, FN = BN / WN / 16;   // 16x16 MFMA tiles per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u / WN, wn = wave_u % WN;

  // ---- XCD-aware, grouped tile mapping (bijective for any grid size) ----
  const int ntile = p.tiles_m * p.tiles_n;
  const int nblk = ntile * p.splitk;
  const int bid = blockIdx.x;
  const int qx = nblk >> 3, rx = nblk & 7, xcd = bid & 7, pos = bid >> 3;
  const int logical_all = (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + pos;
  const int slice = logical_all / ntile;
  const int logical = logical_all - slice * ntile;
  constexpr int GM = 8;
  const int per_group = GM * p.tiles_n;
  const int group = logical / per_group, in_g = logical - group * per_group;
  const int first_m = group * GM;
  const int gsz = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
  const int tm = first_m + in_g % gsz, tn = in_g / gsz;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; i++)
#pragma unroll
    for (int j = 0; j < FN; j++) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk_all = (int)(p.K / BK);
  const int kt0 = slice * p.ktiles_per;
  const int kt1 = (kt0 + p.ktiles_per) < nk_all ? (kt0 + p.ktiles_per) : nk_all;

  // fragment read offsets (bytes inside a tile), constant across K-tiles
  const int frow = lane & 15, fg = lane >> 4;
  int a_off[FM][BK / 32], b_off[FN][BK / 32];
#pragma unroll
  for (int ks = 0; ks < BK / 32; ks++) {
    const int c = ks * 4 + fg;
#pragma unroll
    for (int i = 0; i < FM; i++) {
      const int r_ = wm * (FM * 16) + i * 16 + frow;
      a_off[i][ks] = r_ * (BK * 2) + ((c ^ swz_of<BK>(r_)) * 16);
    }
#pragma unroll
    for (int j = 0; j < FN; j++) {
      const int r_ = wn * (FN * 16) + j * 16 + frow;
      b_off[j][ks] = r_ * (BK * 2) + ((c ^ swz_of<BK>(r_)) * 16);
    }
  }

  constexpr int LOADS = NIT_A + NIT_B;  // LDS-DMA instructions per thread per stage
  const int nkt = kt1 - kt0;
  // prologue: put the first NS-1 stages in flight
#pragma unroll
  for (int st = 0; st < NS - 1; st++) {
    if (st < nkt) {
      char* slot = smem + st * STAGE_BYTES;
      stage_issue<BK, BM, NT>(p.A, p.lda, m0, p.M, (int64_t)(kt0 + st) * BK, slot, tid, wave_u);
      stage_issue<BK, BN, NT>(p.B, p.ldb, n0, p.N, (int64_t)(kt0 + st) * BK, slot + A_BYTES, tid, wave_u);
    }
  }

  for (int it = 0; it < nkt; it++) {
    char* cur = smem + (it % NS) * STAGE_BYTES;
    char* nxt = smem + ((it + NS - 1) % NS) * STAGE_BYTES;
    // stage `it` must have landed; in steady state NS-2 younger stages stay in flight
    if (it + NS - 2 <= nkt - 1) wait_vmcnt<(NS - 2) * LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // every wave's DMA for stage `it` landed; every wave finished reading stage it-1
    if (it + NS - 1 < nkt) {
      stage_issue<BK, BM, NT>(p.A, p.lda, m0, p.M, (int64_t)(kt0 + it + NS - 1) * BK, nxt, tid, wave_u);
      stage_issue<BK, BN, NT>(p.B, p.ldb, n0, p.N, (int64_t)(kt0 + it + NS - 1) * BK, nxt + A_BYTES, tid, wave_u);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 32; ks++) {
      bf16x8_t af[FM], bfr[FN];
#pragma unroll
      for (int j = 0; j < FN; j++) bfr[j] = *(const bf16x8_t*)(cur + A_BYTES + b_off[j][ks]);
#pragma unroll
      for (int i = 0; i < FM; i++) af[i] = *(const bf16x8_t*)(cur + a_off[i][ks]);
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  gemm_epilogue<EPI, FM, FN>(p, acc, m0 + wm * (FM * 16), n0 + wn * (FN * 16), frow, fg, slice);
}

// out[m][n] = alpha * sum_s ws[s][m][n] + beta * out[m][n]   (deterministic split-K combine)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float4* __restrict__ ws, float* __restrict__ out,
                                                            int64_t M, int64_t N, int64_t ldc, int S, float alpha,
                                                            float beta) {
  const int64_t n4 = M * N / 4, nq = N / 4;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    float4 acc = ws[q];
    for (int s = 1; s < S; s++) {
      const float4 v = ws[(int64_t)s * n4 + q];
      acc.x += v.x;
      acc.y += v.y;
      acc.z += v.z;
      acc.w += v.w;
    }
    const int64_t m = q / nq, n = (q - m * nq) * 4;
    float* cp = out + m * ldc + n;
    float4 o = make_float4(acc.x * alpha, acc.y * alpha, acc.z * alpha, acc.w * alpha);
    if (beta != 0.f) {
      const float4 c0 = *(const float4*)cp;
      o.x += beta * c0.x;
      o.y += beta * c0.y;
      o.z += beta * c0.z;
      o.w += beta * c0.w;
    }
    *(float4*)cp = o;
  }
}

template <int BK, int NS, int EPI, int BM, int BN, int WM, int WN>
static int launch_gemm(const GemmArgs& a, void* ws, int64_t ws_bytes, hipStream_t stream) {
  constexpr int smem = NS * (BM + BN) * BK * 2;
  // function-local static with an initialiser: set exactly once, thread-safe (the C ABI is re-entrant)
  static VjPerDeviceOnce attr_once;   // the dynamic-LDS limit is a per-device attribute of the function
  attr_once([] {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<BK, NS, EPI, BM, BN, WM, WN>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  });
  GemmArgs b = a;
  b.tiles_m = (int)cdiv64(a.M, BM);
  b.tiles_n = (int)cdiv64(a.N, BN);
  b.splitk = 1;
  b.ws = nullptr;
  const int nk = (int)(a.K / BK);
  if (EPI == EPI_F32 && ws != nullptr) {
    // wgrad: fill the chip (>= 2 waves of workgroups); each slice keeps >= 8 K-tiles of work
    const int64_t tiles = (int64_t)b.tiles_m * b.tiles_n;
    // concurrent workgroups per round: 2 per CU for the 4-wave tiles, 1 per CU for 256x256; K-tile time scales with
    // the tile area and BK
    const int slots = (BM * BN >= 256 * 256) ? 256 : 512;
    const double us_kt = 1.45 * ((double)BM * BN * BK) / (256.0 * 256.0 * 64.0) * (slots == 512 ? 2.6 : 1.3);
    b.splitk = pick_splitk(tiles, nk, slots, us_kt, 8, a.M, a.N, ws_bytes);
    b.ws = (float*)ws;
  }
  b.ktiles_per = (nk + b.splitk - 1) / b.splitk;
  b.splitk = (nk + b.ktiles_per - 1) / b.ktiles_per;  // no empty slices
  const int nblk = b.tiles_m * b.tiles_n * b.splitk;
  hipLaunchKernelGGL((gemm_nt_kernel<BK, NS, EPI, BM, BN, WM, WN>), dim3(nblk), dim3(WM * WN * 64), smem, stream,
                     b);
  VJ_LAUNCH_CHECK("vj_gemm_bf16_nt");
  if (b.splitk > 1) {
    const int64_t n4 = a.M * a.N / 4;
    int64_t g = cdiv64(n4, 256);
    if (g > 256 * 8) g = 256 * 8;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const float4*)b.ws,
                       (float*)a.C, a.M, a.N, a.ldc, b.splitk, a.alpha, a.beta);
    VJ_LAUNCH_CHECK("vj_gemm_bf16_nt(splitk reduce)");
  }
  return 0;
}

// flags: bits 4-5 = tile config (0 auto, 1 = 128x128, 2 = 256x256); bits 6-7 = pipeline (0 auto, 1 = BK64 double buffer, 2 = BK32 4-stage
//        ring with counted vmcnt, 3 = 256x256 staggered 8-phase schedule, gemm8.hip / gemm8p.hip); bit 8 = the 4-wave 256x128 kernel with two
//        workgroups per CU (gemm4w.hip: single-stream inference).  (Register-staged operands, a 256x128 BK32 ring and a persistent form of
//        the 4-wave kernel existed through round 5; all measured slower in the step: profiles/r03_abab_switches.md, r05_gemm_4wp.md.)
int vj_gemm_launch_8phase(const GemmArgs& a, int epilogue, void* ws, int64_t ws_bytes, hipStream_t stream);  // gemm8.hip
int vj_gemm_launch_4w(const GemmArgs& a, int epilogue, void* ws, int64_t ws_bytes, hipStream_t stream);      // gemm4w.hip
int vj_gemm_launch_8phase_persist(const GemmArgs& a, int epilogue, hipStream_t stream);                        // gemm8p.hip (-100: n/a)

template <int EPI>
static int dispatch_gemm(const GemmArgs& a, int flags, void* ws, int64_t ws_bytes, hipStream_t stream) {   // (flags: by value, edited below)
  int cfg = (flags >> 4) & 3;
  int pipe = (flags >> 6) & 3;
  const bool is_wgrad = (EPI == EPI_F32 && ws != nullptr);
  if (a.lnf_rs != nullptr) flags &= ~0x100;   // the 4-wave kernel carries no folded-LayerNorm epilogue
  // bit 8: the 4-wave 256x128 kernel with two workgroups per CU (gemm4w.hip)
  if ((flags & 0x100) && a.K % 64 == 0) return vj_gemm_launch_4w(a, EPI, ws, ws_bytes, stream);
  if (pipe == 0 && cfg == 0) {
    // measured on the ViT-L step shapes (tools/gemm_bench.py): the staggered 8-phase 256x256 kernel wins on every
    // forward / dgrad shape with >= ~90 tiles; split-K wgrads (few tiles, long K) stay on 128x128 with 2 workgroups
    // per CU.  The 4-wave 256x128 kernel with two workgroups per CU (gemm4w.hip) is FASTER in isolation on most step
    // shapes (+35 % on the K = N = 1024 projection, +32 % predictor qkv, +17 % patch embed / context qkv: the tile-fixed
    // costs hide under the co-resident workgroup) and makes a single-stream step 1.9 % faster, but the training step
    // runs two HIP streams, where the other stream's kernels already fill the 8-phase kernel's idle CUs, and half-CU
    // workgroups then share CUs with attention / LayerNorm workgroups at half of their tuned occupancy: same-box A/B
    // 87.0 ms (8-phase) vs 90.0 ms (4-wave) per step, for every selection policy tried.  So it is opt-in: flags bit 8 or
    // VJ_GEMM_4W=1 (all forward / dgrad GEMMs), meant for single-stream use (inference, profiling).
    const int64_t t4w = cdiv64(a.M, 256) * cdiv64(a.N, 128);
    const int use_4w = a.lnf_rs != nullptr ? 0 : vj_opt(VJ_OPT_GEMM_4W);
    if (use_4w == 1 && !is_wgrad && a.K % 64 == 0 && t4w >= 64) return vj_gemm_launch_4w(a, EPI, ws, ws_bytes, stream);
    const int64_t t256 = cdiv64(a.M, 256) * cdiv64(a.N, 256);
    if (!is_wgrad && a.K % 64 == 0 && t256 >= 90) pipe = 3;
    if (is_wgrad && a.K % 64 == 0 && t256 >= 40) pipe = 3;   // qkv/fc1/fc2 wgrads: 8-phase + split-K (0.97-1.08 vs 0.78-0.96 PF)
  }
  if (cfg == 0) cfg = 1;
  if (pipe == 0) pipe = 1;       // BK64 double buffer (beats the BK32 ring on every step shape)
  if (a.K % 64 != 0) pipe = 2;   // K % 32 only fits the BK32 pipeline
  if (pipe == 3 && a.K % 64 == 0) {
    // persistent variant (gemm8p.hip): one workgroup per CU walks its tiles, next tile's operands prefetched under the
    // current tile, epilogue stores drained under the next K loop; bit-identical outputs.  Run-time option "gemm_persist".
    if (EPI != EPI_F32 && vj_opt(VJ_OPT_GEMM_PERSIST) != 0 && !(a.dbg & 2)) {
      const int rc = vj_gemm_launch_8phase_persist(a, EPI, stream);
      if (rc != -100) return rc;
    }
    if (a.lnf_rs == nullptr) return vj_gemm_launch_8phase(a, EPI, ws, ws_bytes, stream);
    pipe = 1;   // folded LayerNorm: only the persistent kernel and the generic kernels below carry that epilogue
    if (cfg == 1 && a.M >= 256 && a.N >= 256) cfg = 2;
  }
  if (pipe == 3) pipe = 1;
  if (pipe == 1)
    return cfg == 2 ? launch_gemm<64, 2, EPI, 256, 256, 2, 4>(a, ws, ws_bytes, stream)
                    : launch_gemm<64, 2, EPI, 128, 128, 2, 2>(a, ws, ws_bytes, stream);
  return cfg == 2 ? launch_gemm<32, 4, EPI, 256, 256, 2, 4>(a, ws, ws_bytes, stream)
                  : launch_gemm<32, 4, EPI, 128, 128, 2, 2>(a, ws, ws_bytes, stream);
}

static int gemm_entry(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                      int64_t N, int64_t K, const float* bias, const void* residual, int64_t ldr, const void* aux_in,
                      void* aux_out, int64_t ldaux, int epilogue, float alpha, float beta, int flags, void* ws,
                      int64_t ws_bytes, hipStream_t stream, const float* lnf_rs = nullptr, const float* lnf_c = nullptr) {
  VJ_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "vj_gemm_bf16_nt: negative dim");
  if (M == 0 || N == 0) return 0;
  VJ_CHECK_ARG(K > 0 && K % 32 == 0, "vj_gemm_bf16_nt: K=%ld must be a positive multiple of 32 (pad the operands)", (long)K);
  VJ_CHECK_ARG(N % 4 == 0, "vj_gemm_bf16_nt: N=%ld must be a multiple of 4", (long)N);
  VJ_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K,
               "vj_gemm_bf16_nt: lda/ldb must be multiples of 8 and >= K (lda=%ld ldb=%ld K=%ld)", (long)lda,
               (long)ldb, (long)K);
  VJ_CHECK_ARG(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0), "vj_gemm_bf16_nt: A/B must be 16-byte aligned");
  VJ_CHECK_ARG(ldc % 4 == 0 && ldc >= N, "vj_gemm_bf16_nt: ldc=%ld must be a multiple of 4 and >= N", (long)ldc);
  float qscale = 0.f;
  if (epilogue == EPI_QKV_API) {   // qkv projection whose q third carries alpha (= scale * log2 e): bf16 epilogue + column scale
    VJ_CHECK_ARG(N % 12 == 0 && residual == nullptr && alpha != 0.f, "vj_gemm_bf16_nt: epilogue 4 needs N %% 12 == 0, no residual, alpha != 0");
    qscale = alpha;
    epilogue = EPI_BF16;
  }
  VJ_CHECK_ARG(epilogue >= EPI_BF16 && epilogue <= EPI_F32, "vj_gemm_bf16_nt: unknown epilogue %d", epilogue);
  VJ_CHECK_ARG((uintptr_t)C % (epilogue == EPI_F32 ? 16 : 8) == 0, "vj_gemm_bf16_nt: C misaligned");
  if (epilogue == EPI_DGELU) VJ_CHECK_ARG(aux_in != nullptr && ldaux % 4 == 0, "vj_gemm_bf16_nt: EPI_DGELU needs aux_in");
  if (residual) VJ_CHECK_ARG(epilogue == EPI_BF16 && ldr % 4 == 0, "vj_gemm_bf16_nt: residual only with EPI_BF16");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.bias = bias; a.res = (const bf16_t*)residual;
  a.aux_in = (const bf16_t*)aux_in; a.aux_out = (bf16_t*)aux_out;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.ldaux = ldaux;
  a.alpha = alpha; a.beta = beta;
  a.tiles_m = a.tiles_n = 0; a.splitk = 1; a.ktiles_per = 0; a.ws = nullptr;
  a.dbg = vj_opt(VJ_OPT_GEMM_DBG);
  a.zero_row = nullptr;
  a.colpart = nullptr;
  a.raster = 0;
  a.lnf_rs = lnf_rs;
  a.lnf_c = lnf_c;
  a.qscale = qscale;
  a.qcols = qscale != 0.f ? N / 3 : 0;
  switch (epilogue) {
    case EPI_BF16: return dispatch_gemm<EPI_BF16>(a, flags, nullptr, 0, stream);
    case EPI_GELU: return dispatch_gemm<EPI_GELU>(a, flags, nullptr, 0, stream);
    case EPI_DGELU: return dispatch_gemm<EPI_DGELU>(a, flags, nullptr, 0, stream);
    default: return dispatch_gemm<EPI_F32>(a, flags, ws, ws_bytes, stream);
  }
}

extern "C" int vj_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                               int64_t M, int64_t N, int64_t K, const float* bias, const void* residual, int64_t ldr,
                               const void* aux_in, void* aux_out, int64_t ldaux, int epilogue, float alpha,
                               float beta, int flags, hipStream_t stream) {
  return gemm_entry(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, aux_in, aux_out, ldaux, epilogue, alpha, beta,
                    flags, nullptr, 0, stream);
}

// LayerNorm folded into the Linear that consumes it (round 5): C = LayerNorm(X) W^T + b computed from the RAW rows X as
//   C[m,n] = rstd_m * (sum_k X[m,k] Wf[n,k] - mean_m * c[n]) + bf[n],   Wf = bf16(W diag(gamma)), c[n] = sum_k Wf[n,k], bf = b + W beta
// (vj_ln_fold_weights prepares Wf / c / bf, vj_ln_rowstats the rows' {rstd, -mean * rstd}): the LayerNorm output is never written or
// re-read, and the activation is rounded to bf16 once less than on the unfused path.  epilogue: 0 (bf16), 1 (GELU, no saved
// derivative) or 4 (bf16 with the first N/3 columns times alpha: the q third of a qkv projection).  Replaces: norm1 -> attn.qkv and
// norm2 -> mlp.fc1 of a Block whose backward never runs (the EMA target encoder, frozen-encoder inference): modules.py:115,119.
extern "C" int vj_gemm_bf16_nt_lnfold(const void* X, int64_t ldx, const void* Wf, int64_t ldw, void* C, int64_t ldc, int64_t M,
                                      int64_t N, int64_t K, const float* bias_f, const float* rowstats, const float* colsum_w,
                                      int epilogue, float alpha, int flags, hipStream_t stream) {
  VJ_CHECK_ARG(bias_f != nullptr && rowstats != nullptr && colsum_w != nullptr, "vj_gemm_bf16_nt_lnfold: bias_f / rowstats / colsum_w must be given");
  VJ_CHECK_ARG(epilogue == EPI_BF16 || epilogue == EPI_GELU || epilogue == EPI_QKV_API, "vj_gemm_bf16_nt_lnfold: epilogue %d (0, 1 or 4)", epilogue);
  VJ_CHECK_ARG(((uintptr_t)rowstats % 8 == 0) && ((uintptr_t)colsum_w % 16 == 0) && ((uintptr_t)bias_f % 16 == 0),
               "vj_gemm_bf16_nt_lnfold: rowstats / colsum_w / bias_f misaligned");
  return gemm_entry(X, ldx, Wf, ldw, C, ldc, M, N, K, bias_f, nullptr, 0, nullptr, nullptr, 0, epilogue, alpha, 0.0f, flags, nullptr, 0,
                    stream, rowstats, colsum_w);
}

// fc2 dgrad with the bias gradient of fc1 fused: C = (A B^T) * aux_in (EPI_DGELU) and, when the persistent 256x256 kernel
// takes the problem, colpart[2 * cdiv(M,256)][N] = per-(row tile, wave row) fp32 column sums of C before the bf16 rounding
// (*fused = 1; reduce them with vj_reduce_segments).  Otherwise the plain GEMM runs and *fused = 0 (the caller sums C itself).
// C is bit-identical to vj_gemm_bf16_nt's either way.  Replaces: autograd of Mlp.fc1's bias (modules.py:31-34).
extern "C" int64_t vj_gemm_colsum_rows(int64_t M) { return 2 * cdiv64(M, 256); }

extern "C" int vj_gemm_bf16_nt_dgelu_colsum(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                            int64_t M, int64_t N, int64_t K, const void* aux_in, int64_t ldaux,
                                            float* colpart, int64_t colpart_rows, int flags, int* fused,
                                            hipStream_t stream) {
  VJ_CHECK_ARG(fused != nullptr, "vj_gemm_bf16_nt_dgelu_colsum: null `fused`");
  *fused = 0;
  const int64_t t256 = cdiv64(M, 256) * cdiv64(N, 256);
  if (colpart != nullptr && flags == 0 && M > 0 && N > 0 && K > 0 && K % 64 == 0 && t256 >= 90 && vj_opt(VJ_OPT_GEMM_4W) == 0 &&
      vj_opt(VJ_OPT_GEMM_PERSIST) != 0 && vj_opt(VJ_OPT_GEMM_DBG) == 0 && aux_in != nullptr && lda % 8 == 0 && ldb % 8 == 0 &&
      lda >= K && ldb >= K && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ldc % 4 == 0 && ldc >= N &&
      ldaux % 4 == 0 && N % 4 == 0 && (uintptr_t)C % 8 == 0) {
    // the same kernel the automatic selection of vj_gemm_bf16_nt picks for this shape (>= 90 tiles of 256 x 256, K % 64 == 0)
    VJ_CHECK_ARG(colpart_rows >= vj_gemm_colsum_rows(M), "vj_gemm_bf16_nt_dgelu_colsum: colpart has %ld rows, needs %ld",
                 (long)colpart_rows, (long)vj_gemm_colsum_rows(M));
    GemmArgs a;
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.bias = nullptr; a.res = nullptr;
    a.aux_in = (const bf16_t*)aux_in; a.aux_out = nullptr;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = 0; a.ldaux = ldaux;
    a.alpha = 1.0f; a.beta = 0.0f;
    a.tiles_m = a.tiles_n = 0; a.splitk = 1; a.ktiles_per = 0; a.ws = nullptr;
    a.dbg = 0;
    a.zero_row = nullptr;
    a.qscale = 0.f;
    a.qcols = 0;
    a.colpart = colpart;
    a.raster = 0;
    a.lnf_rs = nullptr;
    a.lnf_c = nullptr;
    const int rc = vj_gemm_launch_8phase_persist(a, EPI_DGELU, stream);
    if (rc != -100) {
      *fused = (rc == 0);
      return rc;
    }
  }
  return gemm_entry(A, lda, B, ldb, C, ldc, M, N, K, nullptr, nullptr, 0, aux_in, nullptr, ldaux, EPI_DGELU, 1.0f, 0.0f, flags,
                    nullptr, 0, stream);
}

// wgrad form: C (fp32) = alpha * A B^T + beta * C with the long K (= tokens) dimension split across workgroups
// when the [M,N] tile grid alone cannot fill 256 CUs; partials are combined deterministically from `ws`.
extern "C" int vj_gemm_bf16_nt_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                                      int64_t M, int64_t N, int64_t K, float alpha, float beta, int flags, void* ws,
                                      int64_t ws_bytes, hipStream_t stream) {
  VJ_CHECK_ARG(ws != nullptr && ws_bytes >= M * N * 4, "vj_gemm_bf16_nt_splitk: workspace must hold at least M*N fp32");
  return gemm_entry(A, lda, B, ldb, C, ldc, M, N, K, nullptr, nullptr, 0, nullptr, nullptr, 0, EPI_F32, alpha, beta,
                    flags, ws, ws_bytes, stream);
}
