// 8-phase bf16 MFMA GEMM for gfx950: C[M,N] = A[M,K] * B[N,K]^T, 256x256 tile, 8 waves (2 x 4), BK = 64.
//
// Schedule (two K-tiles = 8 phases per loop trip; all numbers per workgroup):
//   * A K-tile is staged as four 16 KB PARTS: B0, B1, A1 and (for the NEXT K-tile) A0, where A(mq) holds the 64
//     rows of m-fragments 4mq..4mq+3 of BOTH wave rows and B(nq) the 32 rows of n-fragments 2nq,2nq+1 of all four
//     wave columns.  Part q lives in LDS slot q % 8 (8 x 16 KB = 128 KB = two K-tiles).
//   * Phase q has a LOAD section L(q): ds_read_b128 the fragments of part q into one of four register sets
//     (RB0 | RB1 | RA1 | RA0-of-next-tile: 4 or 8 reads), issue the LDS-DMA of part q+4 (2 x global_load_lds_dwordx4
//     per thread), s_waitcnt vmcnt(6) so that part q+1 has landed while parts q+2..q+4 stay in flight ACROSS the
//     barrier; and a COMPUTE section C(q): 16 MFMA 16x16x32 (one quadrant of the wave's 128x64 tile x K=64) under
//     s_setprio 1.  Sections are separated by raw s_barrier (no vmcnt drain).
//   * Waves 0-3 and waves 4-7 (one of each per SIMD) run ONE barrier interval apart, so on every SIMD the LDS/DMA
//     section of one wave overlaps the MFMA section of the other.
// Hazards: part q is read only after a barrier that follows every thread's vmcnt for it (end of L(q-1)); its slot is
// re-filled by part q+8 issued in L(q+4), >= 7 barrier intervals after the last read.
#include "gemm_common.hpp"

#define P8_BM 256
#define P8_BN 256
#define P8_BK 64
#define PART_BYTES 16384

__device__ __forceinline__ void cfence() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ void bar() {
  cfence();
  __builtin_amdgcn_sched_barrier(0);   // nothing (MFMA, ds_read, DMA issue) may be scheduled across a section boundary
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  cfence();
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LDS-DMA through inline assembly (both addressing forms): the compiler then neither tracks these loads in its own
// vmcnt bookkeeping nor assumes LDS was written, so the only waits in the K loop are the counted ones placed by hand.
__device__ __forceinline__ unsigned lds_addr8(const char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void dma16_v(const void* gsrc, unsigned lds_dst) {   // 64-bit per-lane address
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_dst), "v"(gsrc) : "memory");
}
__device__ __forceinline__ void dma16_sv(const char* sbase, unsigned voff, unsigned lds_dst) {   // SGPR base + 32-bit lane offset
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

// Interior tiles (all 256 x 256 rows exist): a part's two chunks per thread are rows r0 and r0 + 64 of the part at the
// same swizzled chunk column, so their addresses are a WAVE-UNIFORM base per (operand, sub-part, j) -- fixed for the
// whole K loop, held in SGPRs -- plus one 32-bit per-thread byte offset per operand (+ 128 bytes per K-tile).  That
// removes the 64-bit VALU address arithmetic (v_mad_u64_u32 / v_lshl_add_u64 per chunk) from every load section.
struct PartBase8 {
  const char* a[2][2];   // [sub-part mq][j]: A rows m0 + j*128 + mq*64 (+ r0)
  const char* b[2][2];   // [sub-part nq][j]: B rows n0 + j*128 + nq*32 (+ (r0>>5)*64 + (r0&31))
  unsigned voff_a, voff_b;
};

// issue the LDS-DMA of part q (q = -1 .. 4*nk-1) into slot (q+8) % 8
template <bool EDGE>
__device__ __forceinline__ void issue_part(const GemmArgs& p, const PartBase8& pb, int q, int64_t m0, int64_t n0, int kt0,
                                           char* smem, int tid, int wave_u) {
  // q = 4t+0 -> B0(t), 4t+1 -> B1(t), 4t+2 -> A1(t), 4t+3 -> A0(t+1);  q = -1 -> A0(0).  With qq = q + 1:
  // qq = 4T + kind, kind 0 = A0(T), 1 = B0(T), 2 = B1(T), 3 = A1(T)
  const int qq = q + 1;
  const int kind = qq & 3;
  const int t = qq >> 2;
  const unsigned slot = lds_addr8(smem) + ((q + 8) & 7) * PART_BYTES;
  const bool isA = (kind == 0) || (kind == 3);
  const int sub = (kind == 0) ? 0 : (kind == 3 ? 1 : kind - 1);  // mq for A parts, nq for B parts
  if constexpr (!EDGE) {
    const unsigned koff = (unsigned)t * (P8_BK * 2);   // bytes along K
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const unsigned dst = slot + (j * 512 + wave_u * 64) * 16;
      if (isA) dma16_sv(pb.a[sub][j], pb.voff_a + koff, dst);
      else dma16_sv(pb.b[sub][j], pb.voff_b + koff, dst);
    }
    return;
  }
  const int64_t k0 = (int64_t)(kt0 + t) * P8_BK;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int c16 = j * 512 + tid;            // 16-byte chunk index inside the part (linear LDS image)
    const int pr = c16 >> 3, cpos = c16 & 7;  // part row, chunk position
    const int c = cpos ^ (pr & 7);            // source-side XOR swizzle (the DMA destination is lane-linear)
    const bf16_t* src;
    if (isA) {
      int64_t gr = m0 + (pr >> 6) * 128 + sub * 64 + (pr & 63);
      gr = gr < p.M ? gr : p.M - 1;
      src = p.A + gr * p.lda + k0 + c * 8;
    } else {
      int64_t gr = n0 + (pr >> 5) * 64 + sub * 32 + (pr & 31);
      gr = gr < p.N ? gr : p.N - 1;
      src = p.B + gr * p.ldb + k0 + c * 8;
    }
    dma16_v(src, slot + (j * 512 + wave_u * 64) * 16);
  }
}

// One workgroup's tile.  EDGE: the tile sticks out of the matrix (per-chunk clamped 64-bit addressing); interior tiles use
// the SGPR-base form.  Two instantiations of the same schedule; the kernel picks one per workgroup (uniform branch).
template <int EPI, bool EDGE>
__device__ __forceinline__ void gemm8_body(const GemmArgs& p, char* smem, int tm, int tn, int slice) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 2, wn = wave_u & 3;
  const bool late_group = wave_u >= 4;        // waves 4-7 run one barrier interval behind waves 0-3
  const int frow = lane & 15, fg = lane >> 4;

  const int64_t m0 = (int64_t)tm * P8_BM, n0 = (int64_t)tn * P8_BN;
  const int nk_all = (int)(p.K / P8_BK);
  const int kt0 = slice * p.ktiles_per;                   // this workgroup's K-tile range (split-K for wgrads)
  const int nk = (kt0 + p.ktiles_per < nk_all ? kt0 + p.ktiles_per : nk_all) - kt0;
  const int last_part = 4 * nk - 2;   // parts: -1 (A0 of tile 0), then per tile B0, B1, A1 and A0 of the next tile
  PartBase8 pb;
  if constexpr (!EDGE) {
    const int r0 = tid >> 3, cpos = tid & 7;          // chunk j of a part: part row j*64 + r0, chunk column cpos
    const int c = cpos ^ (r0 & 7);
    const int rb = (r0 >> 5) * 64 + (r0 & 31);        // B parts interleave the four wave columns' 32-row halves
    pb.voff_a = (unsigned)((r0 * p.lda + c * 8) * 2);
    pb.voff_b = (unsigned)((rb * p.ldb + c * 8) * 2);
    const char* a0 = (const char*)(p.A + m0 * p.lda + (int64_t)kt0 * P8_BK);
    const char* b0 = (const char*)(p.B + n0 * p.ldb + (int64_t)kt0 * P8_BK);
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        pb.a[sub][j] = a0 + (int64_t)(j * 128 + sub * 64) * p.lda * 2;
        pb.b[sub][j] = b0 + (int64_t)(j * 128 + sub * 32) * p.ldb * 2;
      }
  }

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t ra0[4][2], ra1[4][2], rb0[2][2], rb1[2][2];  // [fragment][k-step]

  // per-lane fragment byte offsets inside a part (rows are 128 B, chunks XOR-swizzled by row & 7)
  int a_off[4][2], b_off[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    const int c = ks * 4 + fg;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int pr = wm * 64 + i * 16 + frow;
      a_off[i][ks] = pr * 128 + ((c ^ (pr & 7)) * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int pr = wn * 32 + j * 16 + frow;
      b_off[j][ks] = pr * 128 + ((c ^ (pr & 7)) * 16);
    }
  }

  // ---- prologue: parts -1 (A0 of tile 0), 0, 1, 2 in flight; part -1 landed for everyone
  issue_part<EDGE>(p, pb, -1, m0, n0, kt0, smem, tid, wave_u);
#pragma unroll
  for (int q = 0; q < 3; q++)
    if (q <= last_part) issue_part<EDGE>(p, pb, q, m0, n0, kt0, smem, tid, wave_u);
  if (last_part >= 2) wait_vm<6>();
  else wait_vm<0>();
  bar();
  if (late_group) bar();
  // L(-1): A0 fragments of tile 0; issue part 3; part 0 must have landed before the next barrier
  {
    const char* slot = smem + 7 * PART_BYTES;   // (-1 + 8) % 8
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) ra0[i][ks] = *(const bf16x8_t*)(slot + a_off[i][ks]);
    if (3 <= last_part) issue_part<EDGE>(p, pb, 3, m0, n0, kt0, smem, tid, wave_u);
    if (3 <= last_part) wait_vm<6>();
    else wait_vm<0>();
  }
  bar();
  bar();

  for (int t = 0; t < nk; t++) {
#pragma unroll
    for (int ph = 0; ph < 4; ph++) {
      const int q = 4 * t + ph;
      const char* slot = smem + (q & 7) * PART_BYTES;
      // ---------------- L(q)
#ifdef VJ_GEMM8_SKIP_READS   // timing experiment (A/B build `python -m jepa_amd.build skipreads -DVJ_GEMM8_SKIP_READS`): after the
      if (t > 0) {          // first K-tile the fragment reads are skipped -- WRONG results, same MFMA operands forever -- which
      } else                // bounds what moving the LDS reads out of the load sections could buy
#endif
      if (ph == 0) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int ks = 0; ks < 2; ks++) rb0[j][ks] = *(const bf16x8_t*)(slot + b_off[j][ks]);
      } else if (ph == 1) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int ks = 0; ks < 2; ks++) rb1[j][ks] = *(const bf16x8_t*)(slot + b_off[j][ks]);
      } else if (ph == 2) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int ks = 0; ks < 2; ks++) ra1[i][ks] = *(const bf16x8_t*)(slot + a_off[i][ks]);
      } else if (t + 1 < nk) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int ks = 0; ks < 2; ks++) ra0[i][ks] = *(const bf16x8_t*)(slot + a_off[i][ks]);
      }
      if (q + 4 <= last_part) {
        issue_part<EDGE>(p, pb, q + 4, m0, n0, kt0, smem, tid, wave_u);
        wait_vm<6>();                              // part q+1 landed; q+2..q+4 in flight
      } else {                                     // tail: fewer younger parts behind part q+1
        const int younger = last_part - (q + 1);
        if (younger >= 2) wait_vm<4>();
        else if (younger == 1) wait_vm<2>();
        else wait_vm<0>();
      }
      bar();
      // ---------------- C(q): one quadrant x K = 64
      __builtin_amdgcn_s_setprio(1);
      if (ph == 0) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb0[j][ks], ra0[i][ks], acc[i][j], 0, 0, 0);
      } else if (ph == 1) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
              acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb1[j][ks], ra0[i][ks], acc[i][2 + j], 0, 0, 0);
      } else if (ph == 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
              acc[4 + i][2 + j] =
                  __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb1[j][ks], ra1[i][ks], acc[4 + i][2 + j], 0, 0, 0);
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
              acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb0[j][ks], ra1[i][ks], acc[4 + i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      bar();
    }
  }
  if (!late_group) bar();   // the early group matches the late group's extra barrier
  if (p.dbg & 1) {   // diagnostics: no output traffic (keeps the accumulators alive through one predicated store)
    if (acc[0][0][0] == 12345.678f && acc[7][3][3] == 0.5f) *(float*)p.C = acc[3][2][1];
    return;
  }
  // every wave is past its last LDS read and every DMA has landed: the ring is free, 16 KB of it per wave.
  // The lane id is re-derived here (mbcnt) so that nothing epilogue-only stays live across the K loop (256 VGPRs).
  const int elane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int efrow = elane & 15, efg = elane >> 4;
  if (!(p.dbg & 2) && gemm_epilogue_try_staged<EPI, 8, false>(p, acc, m0 + wm * 128, n0 + wn * 64, efrow, efg, elane,
                                                    smem + wave_u * 16384))
    return;
  gemm_epilogue<EPI, 8, 4, /*INTERIOR_VARIANT=*/(EPI == EPI_F32), false>(p, acc, m0 + wm * 128, n0 + wn * 64, efrow, efg, slice);
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_8phase_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntile = p.tiles_m * p.tiles_n;
  const int logical_all = xcd_logical(blockIdx.x, ntile * p.splitk);
  const int slice = logical_all / ntile;
  int tm, tn;
  tile_of(logical_all - slice * ntile, p.tiles_m, p.tiles_n, tm, tn);
#ifdef VJ_GEMM8_VADDR_ONLY   // A/B build: every tile on the per-chunk 64-bit addressing
  const bool interior = false;
#else
  const bool interior = ((int64_t)(tm + 1) * P8_BM <= p.M) && ((int64_t)(tn + 1) * P8_BN <= p.N);   // workgroup-uniform
#endif
  if (interior) gemm8_body<EPI, false>(p, smem, tm, tn, slice);
  else gemm8_body<EPI, true>(p, smem, tm, tn, slice);
}

__global__ void splitk_reduce_kernel(const float4* ws, float* out, int64_t M, int64_t N, int64_t ldc, int S,
                                     float alpha, float beta);   // gemm.hip

template <int EPI>
static int launch8(const GemmArgs& a, void* ws, int64_t ws_bytes, hipStream_t stream) {
  constexpr int smem = 8 * PART_BYTES;
  // function-local static with an initialiser: set exactly once, thread-safe (the C ABI is re-entrant)
  static VjPerDeviceOnce attr_once;   // the dynamic-LDS limit is a per-device attribute of the function
  attr_once([] {
    (void)hipFuncSetAttribute((const void*)gemm_nt_8phase_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  });
  GemmArgs b = a;
  b.tiles_m = (int)cdiv64(a.M, P8_BM);
  b.tiles_n = (int)cdiv64(a.N, P8_BN);
  b.splitk = 1;
  b.ws = nullptr;
  const int nk = (int)(a.K / P8_BK);
  if (EPI == EPI_F32 && ws != nullptr) {   // wgrad: one workgroup per CU needs >= ~256 of them; >= 8 K-tiles per slice
    const int64_t tiles = (int64_t)b.tiles_m * b.tiles_n;
    b.splitk = pick_splitk(tiles, nk, 256, 1.45, 8, a.M, a.N, ws_bytes);
    b.ws = (float*)ws;
  }
  b.ktiles_per = (nk + b.splitk - 1) / b.splitk;
  b.splitk = (nk + b.ktiles_per - 1) / b.ktiles_per;
  hipLaunchKernelGGL(gemm_nt_8phase_kernel<EPI>, dim3(b.tiles_m * b.tiles_n * b.splitk), dim3(512), smem, stream, b);
  VJ_LAUNCH_CHECK("vj_gemm_bf16_nt(8-phase)");
  if (b.splitk > 1) {
    const int64_t n4 = a.M * a.N / 4;
    int64_t g = cdiv64(n4, 256);
    if (g > 256 * 8) g = 256 * 8;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const float4*)b.ws,
                       (float*)a.C, a.M, a.N, a.ldc, b.splitk, a.alpha, a.beta);
    VJ_LAUNCH_CHECK("vj_gemm_bf16_nt(8-phase splitk reduce)");
  }
  return 0;
}

// entry used by gemm.hip's dispatcher (pipeline 3); requires K % 64 == 0
int vj_gemm_launch_8phase(const GemmArgs& a, int epilogue, void* ws, int64_t ws_bytes, hipStream_t stream) {
  switch (epilogue) {
    case EPI_BF16: return launch8<EPI_BF16>(a, nullptr, 0, stream);
    case EPI_GELU: return launch8<EPI_GELU>(a, nullptr, 0, stream);
    case EPI_DGELU: return launch8<EPI_DGELU>(a, nullptr, 0, stream);
    default: return launch8<EPI_F32>(a, ws, ws_bytes, stream);
  }
}
