// EXPERIMENT (round 4, not on the training path): one-wave-per-SIMD bf16 MFMA GEMM for gfx950,
//   C[M,N] = A[M,K] * B[N,K]^T (+ bias, + residual), 256 x 256 tile, FOUR waves (2 x 2) of 128 x 128, BK = 32, MFMA 32x32x16.
//
// Question it answers: the production kernels (gemm8.hip / gemm8p.hip) put two waves on every SIMD and alternate
// them between a load and a compute section; their K loop runs at 1.27-1.30 us per 64-wide K-tile against 0.98 us of matrix-pipe time
// at the clock the step sustains (77 %), the difference being section boundaries.  MI355X_MICROARCH.md ("one wave per SIMD
// (512-register kernel)") says a single wave that owns a SIMD can hide <= 5 single-issue instructions under every 32-cycle MFMA
// 32x32x16 and reach the pipe floor.  This kernel is that structure in plain HIP:
//   * a wave keeps a 128 x 128 accumulator tile (4 x 4 MFMA blocks, 256 registers) and issues ONLY back-to-back MFMAs; the fragment
//     reads of the NEXT 16-wide k-step (8 ds_read_b128), the LDS-DMA of the K-tile three ahead (8 instructions per thread and K-tile)
//     and one barrier per K-tile ride in the gaps (about 1.5 instructions per MFMA);
//   * four 32 KB stages (A 256 x 32 | B 256 x 32, 64-byte rows, 16-byte chunks XOR-swizzled by (row >> 2) & 3 so that the 16-lane groups
//     of a ds_read_b128 cover all 64 banks); stage kt + 3 is requested at the top of K-tile kt, right after the barrier that proves
//     every wave is done with stage kt - 1; the same barrier publishes stage kt + 1, whose first fragments are read during the second
//     k-step of K-tile kt;
//   * operands swapped (D = Bfrag x Afrag): lane l owns row l % 32 of a 32 x 32 block and the columns 4 (l >> 5) + 8 q + {0..3}.
// The epilogue is the plain one (bias, residual, bf16 stores straight from the MFMA layout): it exists to check results; the
// measurement is the SLOPE of time over K (tools/gemm_ksweep.py --kernel 1w), which does not contain it.
#include <type_traits>
#include "gemm_common.hpp"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define W1_STAGE 32768
#define W1_NSTAGE 4

namespace {

__device__ __forceinline__ void w1_bar() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void w1_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ unsigned w1_lds(const char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void w1_dma(const char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

__global__ __launch_bounds__(256, 1) void gemm_nt_1w_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // one tile per workgroup, XCD-aware grouped order; edge tiles shifted inside the matrix (overlap written twice, same bits)
  int tm, tn;
  tile_of(xcd_logical(blockIdx.x, p.tiles_m * p.tiles_n), p.tiles_m, p.tiles_n, tm, tn);
  int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
  m0 = m0 + 256 <= p.M ? m0 : p.M - 256;
  n0 = n0 + 256 <= p.N ? n0 : p.N - 256;
  const int nk = (int)(p.K / 32);

  // LDS-DMA: a sweep = 256 threads x 16 bytes = 64 rows x 64 bytes; thread -> row (tid >> 2) of the sweep, PHYSICAL chunk tid & 3,
  // which holds the logical chunk (tid & 3) ^ ((row >> 2) & 3)
  const int drow = tid >> 2, dpc = tid & 3, dlc = dpc ^ ((drow >> 2) & 3);
  const unsigned voff_a = (unsigned)((drow * p.lda + dlc * 8) * 2);
  const unsigned voff_b = (unsigned)((drow * p.ldb + dlc * 8) * 2);
  const char* abase[4];
  const char* bbase[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    abase[s] = (const char*)(p.A + (m0 + s * 64) * p.lda);
    bbase[s] = (const char*)(p.B + (n0 + s * 64) * p.ldb);
  }
  const unsigned ring = w1_lds(smem) + (unsigned)(wave_u * 1024);
  // issue K-tile `kt` into stage kt % 4.  Beyond the last K-tile the LAST one is requested again (into a stage nobody reads any more):
  // every iteration then issues the same eight instructions and the counted waits need no tail cases
  auto issue = [&](int kt) __attribute__((always_inline)) {
    const unsigned st = ring + (unsigned)(kt & (W1_NSTAGE - 1)) * W1_STAGE;
    const unsigned koff = (unsigned)(kt < nk ? kt : nk - 1) * 64u;
#pragma unroll
    for (int s = 0; s < 4; s++) w1_dma(abase[s], voff_a + koff, st + s * 4096);
#pragma unroll
    for (int s = 0; s < 4; s++) w1_dma(bbase[s], voff_b + koff, st + 16384 + s * 4096);
  };

  // fragment byte offsets inside a stage: row r (64-byte rows), logical chunk 2 ks + (lane >> 5), physical = logical ^ ((r >> 2) & 3)
  int a_off[2], b_off[2];
  {
    const int key = (l31 >> 2) & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int pc = (2 * ks + lh) ^ key;
      a_off[ks] = (wm * 128 + l31) * 64 + pc * 16;
      b_off[ks] = 16384 + (wn * 128 + l31) * 64 + pc * 16;
    }
  }

  f32x16_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  bf16x8_t fa[2][4], fb[2][4];   // [register set][32-row block]

  auto read_frags = [&](int set, const char* stage, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; i++) fa[set][i] = *(const bf16x8_t*)(stage + a_off[ks] + i * 2048);
#pragma unroll
    for (int j = 0; j < 4; j++) fb[set][j] = *(const bf16x8_t*)(stage + b_off[ks] + j * 2048);
  };
  auto mma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
  };

  // One K-tile = eight CHUNKS of four MFMAs (one 32-row block x the four column blocks); every chunk carries two fragment reads of
  // the next k-step and ONE LDS-DMA instruction of K-tile kt + 3, pinned between its MFMAs with scheduling barriers: issued in a
  // burst after the barrier, the eight DMA instructions (s_mov m0 + s_nop + global_load_lds each) would leave the matrix pipe idle
  // for their whole issue time -- there is no second wave on the SIMD to cover it.
  //   chunk c = 0..3: k-step a (register set 0), reads of k-step b into set 1;  c = 4..7: k-step b (set 1), reads of K-tile kt + 1's
  //   k-step a into set 0.  Read order: the four B fragments first (every chunk of the next k-step needs them), then the A fragments.
  auto chunk = [&](auto c_tag, int kt, const char* st_cur, const char* st_nxt) __attribute__((always_inline)) {
    constexpr int C = decltype(c_tag)::value;
    constexpr int SET = C >> 2, I = C & 3, OTHER = SET ^ 1;
    const char* src = SET == 0 ? st_cur : st_nxt;   // where the OTHER set's fragments come from
    constexpr int KS = SET == 0 ? 1 : 0;
    const unsigned st = ring + (unsigned)((kt + 3) & (W1_NSTAGE - 1)) * W1_STAGE;
    const unsigned koff = (unsigned)(kt + 3 < nk ? kt + 3 : nk - 1) * 64u;
    acc[I][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[SET][0], fa[SET][I], acc[I][0], 0, 0, 0);
    if constexpr (I < 2) fb[OTHER][2 * I] = *(const bf16x8_t*)(src + b_off[KS] + (2 * I) * 2048);
    else fa[OTHER][2 * (I - 2)] = *(const bf16x8_t*)(src + a_off[KS] + (2 * (I - 2)) * 2048);
    __builtin_amdgcn_sched_barrier(0);
    acc[I][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[SET][1], fa[SET][I], acc[I][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (C < 4) w1_dma(abase[C], voff_a + koff, st + C * 4096);
    else w1_dma(bbase[C - 4], voff_b + koff, st + 16384 + (C - 4) * 4096);
    __builtin_amdgcn_sched_barrier(0);
    acc[I][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[SET][2], fa[SET][I], acc[I][2], 0, 0, 0);
    if constexpr (I < 2) fb[OTHER][2 * I + 1] = *(const bf16x8_t*)(src + b_off[KS] + (2 * I + 1) * 2048);
    else fa[OTHER][2 * (I - 2) + 1] = *(const bf16x8_t*)(src + a_off[KS] + (2 * (I - 2) + 1) * 2048);
    __builtin_amdgcn_sched_barrier(0);
    acc[I][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[SET][3], fa[SET][I], acc[I][3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>;
  using C4 = std::integral_constant<int, 4>;
  using C5 = std::integral_constant<int, 5>;
  using C6 = std::integral_constant<int, 6>;
  using C7 = std::integral_constant<int, 7>;

  // prologue: three K-tiles requested, the first one landed for everybody, its first fragments read
  issue(0);
  issue(1);
  issue(2);
  w1_wait<16>();
  w1_bar();
  read_frags(0, smem, 0);

  for (int kt = 0; kt < nk; kt++) {
    const char* st_cur = smem + (kt & (W1_NSTAGE - 1)) * W1_STAGE;
    const char* st_nxt = smem + ((kt + 1) & (W1_NSTAGE - 1)) * W1_STAGE;
    // K-tile kt + 1 must have landed (it is read in the second half of this iteration); kt + 2 stays in flight
    w1_wait<8>();
    w1_bar();   // publishes stage kt + 1; every wave is done with stage kt - 1 (K-tile kt + 3 goes there)
    chunk(C0{}, kt, st_cur, st_nxt);
    chunk(C1{}, kt, st_cur, st_nxt);
    chunk(C2{}, kt, st_cur, st_nxt);
    chunk(C3{}, kt, st_cur, st_nxt);
    chunk(C4{}, kt, st_cur, st_nxt);
    chunk(C5{}, kt, st_cur, st_nxt);
    chunk(C6{}, kt, st_cur, st_nxt);
    chunk(C7{}, kt, st_cur, st_nxt);
  }

  w1_wait<0>();   // the re-requested tail tiles must have landed before the workgroup may leave
  // plain epilogue straight from the MFMA layout: lane owns row l31 of each 32-row block, columns 4 lh + 8 q + {0..3}
  const bool has_bias = p.bias != nullptr, has_res = p.res != nullptr;
  const int64_t row_l = m0 + wm * 128 + l31, col_l = n0 + wn * 128 + 4 * lh;
  bf16_t* cp = (bf16_t*)p.C + row_l * p.ldc + col_l;
  const bf16_t* rp = has_res ? p.res + row_l * p.ldr + col_l : nullptr;
  const float* bp = has_bias ? p.bias + col_l : nullptr;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      asm volatile("" ::: "memory");   // one 32 x 32 block at a time: keeps the loads / address arithmetic of 64 stores out of flight
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int cofs = j * 32 + 8 * q;
        float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        if (has_bias) {
          const float4 b4 = *(const float4*)(bp + cofs);
          v[0] += b4.x;
          v[1] += b4.y;
          v[2] += b4.z;
          v[3] += b4.w;
        }
        if (has_res) {
          const u32x2_t r2 = *(const u32x2_t*)(rp + cofs);
          v[0] += bf_lo(r2[0]);
          v[1] += bf_hi(r2[0]);
          v[2] += bf_lo(r2[1]);
          v[3] += bf_hi(r2[1]);
        }
        u32x2_t o;
        o[0] = pack_bf2(v[0], v[1]);
        o[1] = pack_bf2(v[2], v[3]);
        if (!(p.dbg & 1)) *(u32x2_t*)(cp + cofs) = o;
      }
    }
    cp += 32 * p.ldc;
    if (has_res) rp += 32 * p.ldr;
  }
  if (p.dbg & 1) {   // K-sweep diagnostics: no output traffic, accumulators kept alive
    if (acc[0][0][0] == 12345.678f && acc[3][3][15] == 0.5f) *(float*)p.C = acc[1][2][3];
  }
}

}  // namespace

// EXPERIMENTAL entry (tools / tests only; the training path never calls it): C (bf16) = A B^T + bias + residual.
// M, N >= 256, K a positive multiple of 32, lda / ldb multiples of 8, N / ldc / ldr multiples of 4, 16-byte-aligned A / B.
extern "C" int vj_gemm_bf16_nt_1w(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                                  int64_t K, const float* bias, const void* residual, int64_t ldr, int dbg, hipStream_t stream) {
  VJ_CHECK_ARG(M >= 256 && N >= 256 && K >= 32 && K % 32 == 0, "vj_gemm_bf16_nt_1w: needs M, N >= 256 and K %% 32 == 0");
  VJ_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && lda < (1 << 23) && ldb < (1 << 23) && N % 4 == 0 &&
                   ldc % 4 == 0 && ldc >= N && (residual == nullptr || (ldr % 4 == 0 && ldr >= N)),
               "vj_gemm_bf16_nt_1w: leading dimensions");
  VJ_CHECK_ARG(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 8 == 0), "vj_gemm_bf16_nt_1w: alignment");
  static VjPerDeviceOnce attr_once;
  attr_once([] {
    (void)hipFuncSetAttribute((const void*)gemm_nt_1w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W1_NSTAGE * W1_STAGE);
  });
  GemmArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.bias = bias; a.res = (const bf16_t*)residual;
  a.aux_in = nullptr; a.aux_out = nullptr;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.ldaux = 0;
  a.alpha = 1.f; a.beta = 0.f;
  a.tiles_m = (int)cdiv64(M, 256); a.tiles_n = (int)cdiv64(N, 256);
  a.splitk = 1; a.ktiles_per = (int)(K / 32); a.ws = nullptr; a.zero_row = nullptr; a.dbg = dbg;
  a.qscale = 0.f; a.qcols = 0; a.colpart = nullptr; a.gelu_lp = 0;
  hipLaunchKernelGGL(gemm_nt_1w_kernel, dim3(a.tiles_m * a.tiles_n), dim3(256), W1_NSTAGE * W1_STAGE, stream, a);
  VJ_LAUNCH_CHECK("vj_gemm_bf16_nt_1w");
  return 0;
}
