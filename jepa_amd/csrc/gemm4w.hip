// Two-workgroups-per-CU bf16 MFMA GEMM for gfx950: C[M,N] = A[M,K] * B[N,K]^T, 256x128 tile, 4 waves (2 x 2), BK = 64.
//
// Why a second schedule next to gemm8.hip: the 8-wave 256x256 kernel owns a whole CU (8 x 256 VGPRs, 128 KB LDS), so
// the ~6.7 us a tile spends outside its K loop (first-DMA latency, epilogue stores) is time the CU's matrix pipes sit
// idle: 22 % of a K = 1024 tile, 43 % of a K = 384 one (profiles/r02_gemm_shapes_vitl16.md).  Here a workgroup is ONE
// wave per SIMD with the same 128x64 wave tile and the same phase structure, and TWO workgroups share a CU (2 x 80 KB
// LDS, 2 x 4 x 256 VGPRs): they are not synchronised with each other, so one workgroup's prologue / epilogue and
// LDS-load sections run under the other's MFMA sections -- the role the second wave group plays inside gemm8.hip -- and
// the tile grid is twice as fine (wave quantisation of the N = 1024 shapes).
//
// Schedule (4 phases per K-tile, one barrier per phase; all numbers per workgroup):
//   * a K-tile is staged as three 16 KB PARTS: B (128 n-rows: the 4 n-fragments of both wave columns), A1 (m-fragments
//     4..7 of both wave rows) and, for the NEXT K-tile, A0 (m-fragments 0..3).  Part p lives in slot (p + 5) % 5 of a
//     5 x 16 KB ring; parts are issued three ahead (one K-tile) and waited for with counted vmcnt (4 DMA instructions
//     per thread and part), never 0 inside the loop.
//   * phase 0: read rb0 (n-fragments 0,1) from B, issue part +3        | 16 MFMA  acc[0..3][0..1] += rb0 x ra0
//     phase 1: read rb1 (n-fragments 2,3) from B, wait for A1          | 16 MFMA  acc[0..3][2..3] += rb1 x ra0
//     phase 2: read ra1 from A1, issue part +3, wait for A0(next)      | 16 MFMA  acc[4..7][2..3] += rb1 x ra1
//     phase 3: read ra0(next) from A0, issue part +3, wait for B(next) | 16 MFMA  acc[4..7][0..1] += rb0 x ra1
// Hazards: a part is read only after a barrier that follows every thread's counted wait for it; a slot is re-filled at
// least two barriers after the section that read it last (see the slot arithmetic next to issue_part4).
#include "gemm_common.hpp"
#include <type_traits>
#include <cstdlib>

#define W4_BM 256
#define W4_BN 128
#define W4_BK 64
#define W4_PART 16384
#define W4_SLOTS 5

__device__ __forceinline__ void bar4() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
}

// wait until at most `parts` younger parts (4 DMA instructions each) are still in flight
__device__ __forceinline__ void wait_parts(int parts) {
  if (parts >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (parts == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// part index p: -1 = A0(0);  3T = B(T), 3T+1 = A1(T), 3T+2 = A0(T+1).  Slot (p+5) % 5: part p+5 overwrites part p, and
// part p+5 is issued in the section that reads part p+2 -- part p was read three sections (>= 2 barriers) earlier.
// A part is 128 rows x 128 bytes = 1024 16-byte chunks; thread `tid` moves chunks tid, tid+256, tid+512, tid+768, i.e.
// rows r0, r0+32, r0+64, r0+96 (r0 = tid >> 3) at the same (swizzled) chunk column.  Interior tiles address them as
// WAVE-UNIFORM row-group base (SGPR pair, fixed for the whole K loop) + one 32-bit per-thread byte offset
// (row r0, swizzled chunk, + 128 bytes per K-tile): the saddr form of global_load_lds_dwordx4.  No 64-bit VALU address
// arithmetic and two address VGPRs in total; issued through inline assembly so that the compiler, which cannot see
// that LDS is written behind its back, also adds no vmcnt(0) of its own in front of later LDS reads.
struct PartBase {
  const char* a[8];   // A0 chunks j = 0..3 (wave row j >> 1, rows +32 (j & 1)), then the same for A1 (+64 rows)
  const char* b[4];   // B chunks j = 0..3 (rows +32 j)
};
__device__ __forceinline__ unsigned lds_addr4(const char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void dma16_s(const char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

// interior tiles: all 256 x 128 rows exist.  kind 0: A0, 1: B, 2: A1.  voff already includes the K-tile's 128 bytes.
template <int KIND>
__device__ __forceinline__ void issue_fast(const PartBase& s, unsigned voff, unsigned lds_slot, int wave_u) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const char* base = KIND == 1 ? s.b[j] : s.a[(KIND == 2 ? 4 : 0) + j];
    dma16_s(base, voff, lds_slot + j * 4096 + wave_u * 1024);   // chunk (j * 256 + wave * 64 + lane) * 16 bytes
  }
}

// edge tiles: rows beyond M / N re-read the last valid row (their products land in rows / columns that are never stored)
__device__ __forceinline__ void issue_edge(const GemmArgs& p, int kind, int tile, int64_t m0, int64_t n0, int kt0,
                                           char* slot, int tid, int wave_u) {
  const int64_t k0 = (int64_t)(kt0 + tile) * W4_BK;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c16 = j * 256 + tid;            // 16-byte chunk index inside the part (linear LDS image)
    const int pr = c16 >> 3, cpos = c16 & 7;  // part row (0..127), chunk position
    const int c = cpos ^ (pr & 7);            // source-side XOR swizzle (the DMA destination is lane-linear)
    const bf16_t* src;
    if (kind != 1) {
      int64_t gr = m0 + (pr >> 6) * 128 + (kind == 2 ? 64 : 0) + (pr & 63);
      gr = gr < p.M ? gr : p.M - 1;
      src = p.A + gr * p.lda + k0 + c * 8;
    } else {
      int64_t gr = n0 + pr;                   // wave column (pr >> 6) * 64 + fragment row (pr & 63)
      gr = gr < p.N ? gr : p.N - 1;
      src = p.B + gr * p.ldb + k0 + c * 8;
    }
    char* dst = slot + (j * 256 + wave_u * 64) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

// K loop of one workgroup.  EDGE: the tile sticks out of the matrix (clamped per-chunk addressing); otherwise one base
// pointer per operand.  The steady state (all three look-ahead parts exist) is straight-line: unconditional issues and a
// constant vmcnt(8); the last two K-tiles run the same phases with existence checks and shrinking wait counts.
template <bool EDGE>
__device__ __forceinline__ void k_loop_4w(const GemmArgs& p, char* smem, f32x4_t (&acc)[8][4], int64_t m0, int64_t n0,
                                          int kt0, int nk, int tid, int wave_u, int wm, int wn, int frow, int fg) {
  const int last_part = 3 * nk - 2;   // A1 of the last K-tile
  bf16x8_t ra0[4][2], ra1[4][2], rb0[2][2], rb1[2][2];  // [fragment][k-step]
  // per-lane fragment byte offsets inside a part (128-byte rows, chunks swizzled by row & 7): fragment i of a wave sits
  // 16 rows = 2048 bytes further with the same swizzle key, so two bases per operand + immediates cover all of them
  int a_base[2], b_base[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    const int c = ks * 4 + fg;
    a_base[ks] = (wm * 64 + frow) * 128 + ((c ^ (frow & 7)) * 16);
    b_base[ks] = (wn * 64 + frow) * 128 + ((c ^ (frow & 7)) * 16);
  }
  PartBase src;
  unsigned voff_a = 0, voff_b = 0;
  if constexpr (!EDGE) {
    const int r0 = tid >> 3, cpos = tid & 7;
    const int c = cpos ^ (r0 & 7);   // rows r0 + 32 j share r0 & 7
    voff_a = (unsigned)((r0 * p.lda + c * 8) * 2);
    voff_b = (unsigned)((r0 * p.ldb + c * 8) * 2);
    const char* a0 = (const char*)(p.A + m0 * p.lda + (int64_t)kt0 * W4_BK);
    const char* b0 = (const char*)(p.B + n0 * p.ldb + (int64_t)kt0 * W4_BK);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      src.a[j] = a0 + ((j >> 1) * 128 + (j & 1) * 32) * p.lda * 2;
      src.a[4 + j] = a0 + ((j >> 1) * 128 + (j & 1) * 32 + 64) * p.lda * 2;
      src.b[j] = b0 + (j * 32) * p.ldb * 2;
    }
  }
  const unsigned lds0 = lds_addr4(smem);
  // kind: 0 A0, 1 B, 2 A1 ; tile = its K-tile ; sl = its ring slot
  auto issue = [&](auto kind_tag, int tile, int sl) {
    constexpr int KIND = decltype(kind_tag)::value;
    if constexpr (EDGE) issue_edge(p, KIND, tile, m0, n0, kt0, smem + sl * W4_PART, tid, wave_u);
    else issue_fast<KIND>(src, (KIND == 1 ? voff_b : voff_a) + (unsigned)tile * 128u, lds0 + sl * W4_PART, wave_u);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  int issued = -2;   // highest part index issued so far (only consulted by the tail iterations)
  auto issue_if = [&](int q, auto kind_tag, int tile, int sl) {
    if (q <= last_part) {
      issue(kind_tag, tile, sl);
      issued = q;
    }
  };
  auto wait_for = [&](int q) {   // this thread's share of part q has landed; younger parts stay in flight
    if (q <= last_part) wait_parts(issued - q);
  };

  // ---- prologue: A0(0) -> slot 4, B(0) -> 0, A1(0) -> 1 in flight; A0(0) and B(0) landed for everyone; A0(1) -> slot 2
  issue_if(-1, K0{}, 0, 4);
  issue_if(0, K1{}, 0, 0);
  issue_if(1, K2{}, 0, 1);
  wait_for(0);
  bar4();
  {
    const char* slot = smem + 4 * W4_PART;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) ra0[i][ks] = *(const bf16x8_t*)(slot + a_base[ks] + i * 2048);
    issue_if(2, K0{}, 1, 2);
  }

  int s0 = 0;   // ring slot of part 3t (B(t)); parts 3t+1 .. 3t+5 follow cyclically
  auto slot_of = [&](int d) {   // slot of part 3t + d, d = 0..5
    int s = s0 + d;
    return s >= 2 * W4_SLOTS ? s - 2 * W4_SLOTS : (s >= W4_SLOTS ? s - W4_SLOTS : s);
  };
  auto mma = [&](auto quad_tag, const bf16x8_t (&rb)[2][2], const bf16x8_t (&ra)[4][2]) {
    constexpr int QI = decltype(quad_tag)::value >> 1, QJ = decltype(quad_tag)::value & 1;   // accumulator quadrant
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[4 * QI + i][2 * QJ + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[j][ks], ra[i][ks], acc[4 * QI + i][2 * QJ + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  using Q00 = std::integral_constant<int, 0>;
  using Q01 = std::integral_constant<int, 1>;
  using Q10 = std::integral_constant<int, 2>;
  using Q11 = std::integral_constant<int, 3>;

  auto k_tile = [&](int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int pb = 3 * t;
    // ---------------- phase 0: rb0 <- B(t); part +3 = B(t+1)
    {
      const char* slot = smem + s0 * W4_PART;
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) rb0[j][ks] = *(const bf16x8_t*)(slot + b_base[ks] + j * 2048);
      if constexpr (TAIL) issue_if(pb + 3, K1{}, t + 1, slot_of(3));
      else issue(K1{}, t + 1, slot_of(3));
      bar4();
      mma(Q00{}, rb0, ra0);
    }
    // ---------------- phase 1: rb1 <- B(t); A1(t) must land
    {
      const char* slot = smem + s0 * W4_PART;
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) rb1[j][ks] = *(const bf16x8_t*)(slot + b_base[ks] + (2 + j) * 2048);
      if constexpr (TAIL) wait_for(pb + 1);
      else wait_parts(2);
      bar4();
      mma(Q01{}, rb1, ra0);
    }
    // ---------------- phase 2: ra1 <- A1(t); part +3 = A1(t+1); A0(t+1) must land
    {
      const char* slot = smem + slot_of(1) * W4_PART;
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) ra1[i][ks] = *(const bf16x8_t*)(slot + a_base[ks] + i * 2048);
      if constexpr (TAIL) {
        issue_if(pb + 4, K2{}, t + 1, slot_of(4));
        wait_for(pb + 2);
      } else {
        issue(K2{}, t + 1, slot_of(4));
        wait_parts(2);
      }
      bar4();
      mma(Q11{}, rb1, ra1);
    }
    // ---------------- phase 3: ra0 <- A0(t+1); part +3 = A0(t+2); B(t+1) must land
    {
      if (!TAIL || t + 1 < nk) {
        const char* slot = smem + slot_of(2) * W4_PART;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int ks = 0; ks < 2; ks++) ra0[i][ks] = *(const bf16x8_t*)(slot + a_base[ks] + i * 2048);
      }
      if constexpr (TAIL) {
        issue_if(pb + 5, K0{}, t + 2, slot_of(5));
        wait_for(pb + 3);
      } else {
        issue(K0{}, t + 2, slot_of(5));
        wait_parts(2);
      }
      bar4();
      mma(Q10{}, rb0, ra1);
    }
    s0 = slot_of(3);
  };
  int t = 0;
  for (; t < nk - 2; t++) k_tile(t, std::false_type{});
  if (nk >= 3) issued = 3 * nk - 4;   // what the steady state left in flight: parts up to 3(nk-3)+5
  for (; t < nk; t++) k_tile(t, std::true_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bar4();   // every wave is past its last LDS read and every DMA has landed: the ring is free, 16 KB of it per wave
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_4w_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int frow = lane & 15, fg = lane >> 4;

  const int ntile = p.tiles_m * p.tiles_n;
  const int logical_all = xcd_logical(blockIdx.x, ntile * p.splitk);
  const int slice = logical_all / ntile;
  int tm, tn;
  tile_of(logical_all - slice * ntile, p.tiles_m, p.tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * W4_BM, n0 = (int64_t)tn * W4_BN;
  const int nk_all = (int)(p.K / W4_BK);
  const int kt0 = slice * p.ktiles_per;
  const int nk = (kt0 + p.ktiles_per < nk_all ? kt0 + p.ktiles_per : nk_all) - kt0;
  const bool interior = (m0 + W4_BM <= p.M) && (n0 + W4_BN <= p.N);   // workgroup-uniform

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  if (interior) k_loop_4w<false>(p, smem, acc, m0, n0, kt0, nk, tid, wave_u, wm, wn, frow, fg);
  else k_loop_4w<true>(p, smem, acc, m0, n0, kt0, nk, tid, wave_u, wm, wn, frow, fg);
  if (p.dbg & 1) {
    if (acc[0][0][0] == 12345.678f && acc[7][3][3] == 0.5f) *(float*)p.C = acc[3][2][1];
    return;
  }
  const int elane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int efrow = elane & 15, efg = elane >> 4;
  if (!(p.dbg & 2) && gemm_epilogue_try_staged<EPI, 8, false>(p, acc, m0 + wm * 128, n0 + wn * 64, efrow, efg, elane,
                                                              smem + wave_u * 16384))
    return;
  gemm_epilogue<EPI, 8, 4, /*INTERIOR_VARIANT=*/(EPI == EPI_F32), false>(p, acc, m0 + wm * 128, n0 + wn * 64, efrow, efg, slice);
}

// (A persistent form of this kernel -- two workgroups per CU walking tile lists -- was built in round 5: bit-identical, slower on every
//  shape and policy, profiles/r05_gemm_4wp.md.)

__global__ void splitk_reduce_kernel(const float4* ws, float* out, int64_t M, int64_t N, int64_t ldc, int S,
                                     float alpha, float beta);   // gemm.hip

template <int EPI>
static int launch4w(const GemmArgs& a, void* ws, int64_t ws_bytes, hipStream_t stream) {
  constexpr int smem = W4_SLOTS * W4_PART;
  static VjPerDeviceOnce attr_once;   // the dynamic-LDS limit is a per-device attribute of the function
  attr_once([] {
    (void)hipFuncSetAttribute((const void*)gemm_nt_4w_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  });
  static const bool occ_dbg = [] {
    if (getenv("VJ_GEMM_DBG_OCC")) {
      int n = -1;
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)gemm_nt_4w_kernel<EPI>, 256, smem);
      fprintf(stderr, "[vj] gemm_nt_4w_kernel<%d>: %d workgroups per CU (query rc %d, %d B dynamic LDS)\n", EPI, n, (int)e, smem);
    }
    return true;
  }();
  (void)occ_dbg;
  GemmArgs b = a;
  b.tiles_m = (int)cdiv64(a.M, W4_BM);
  b.tiles_n = (int)cdiv64(a.N, W4_BN);
  b.splitk = 1;
  b.ws = nullptr;
  const int nk = (int)(a.K / W4_BK);
  if (EPI == EPI_F32 && ws != nullptr) {   // wgrad: 512 workgroup slots per round, K-tile time of a half-size tile
    const int64_t tiles = (int64_t)b.tiles_m * b.tiles_n;
    b.splitk = pick_splitk(tiles, nk, 512, 1.45, 8, a.M, a.N, ws_bytes);
    b.ws = (float*)ws;
  }
  b.ktiles_per = (nk + b.splitk - 1) / b.splitk;
  b.splitk = (nk + b.ktiles_per - 1) / b.ktiles_per;
  hipLaunchKernelGGL(gemm_nt_4w_kernel<EPI>, dim3(b.tiles_m * b.tiles_n * b.splitk), dim3(256), smem, stream, b);
  VJ_LAUNCH_CHECK("vj_gemm_bf16_nt(4-wave)");
  if (b.splitk > 1) {
    const int64_t n4 = a.M * a.N / 4;
    int64_t g = cdiv64(n4, 256);
    if (g > 256 * 8) g = 256 * 8;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const float4*)b.ws,
                       (float*)a.C, a.M, a.N, a.ldc, b.splitk, a.alpha, a.beta);
    VJ_LAUNCH_CHECK("vj_gemm_bf16_nt(4-wave splitk reduce)");
  }
  return 0;
}

int vj_gemm_launch_4w(const GemmArgs& a, int epilogue, void* ws, int64_t ws_bytes, hipStream_t stream) {
  switch (epilogue) {
    case EPI_BF16: return launch4w<EPI_BF16>(a, nullptr, 0, stream);
    case EPI_GELU: return launch4w<EPI_GELU>(a, nullptr, 0, stream);
    case EPI_DGELU: return launch4w<EPI_DGELU>(a, nullptr, 0, stream);
    default: return launch4w<EPI_F32>(a, ws, ws_bytes, stream);
  }
}
