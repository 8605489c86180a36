// Persistent 8-phase bf16 MFMA GEMM for gfx950: C[M,N] = A[M,K] * B[N,K]^T, 256x256 tiles, 8 waves, BK = 64.
//
// Same K loop as gemm8.hip (four 16 KB parts per K-tile in an 8-slot LDS ring, LDS-DMA four parts ahead, counted vmcnt across
// raw barriers, two wave groups one barrier interval apart, MFMA 16x16x32 with swapped operands) and the same fused epilogues
// (gemm_common.hpp), so every output is BIT-IDENTICAL to the one-tile-per-workgroup kernel.  What changes is everything
// around the K loop.  Measured on the one-tile kernel (profiles/r03_gemm_ksweep_epilogue.md): of the ~9 us a tile round
// spends outside its K loop, ~1.8 us is launch + first-DMA latency and 4-8 us is the epilogue -- not its arithmetic but the
// drain of the stores: the 256 workgroups of a round run in lockstep, dump 256 x 128 KB = 32 MB at once, the write path
// drains that at ~4.4 TB/s and every wave sits in s_endpgm until its stores are acknowledged.  Here:
//
//   * ONE workgroup per CU walks a list of tiles (grid = #CUs).  The part stream is continuous across tiles: the last
//     K-tile of tile i issues the LDS-DMA of tile i+1's parts -1 .. 5 (its first 1.5 K-tiles, into ring slots that are dead
//     by then), part 6 follows right after the epilogue, so the next K loop starts on operands that are already in LDS.
//   * The epilogue's stores are issued and NOT waited for: the next tile computes for five sections (~1.7 us) before its
//     first counted vmcnt, and that wait is merely conservative while stores are still in flight (vmcnt counts stores on
//     gfx9; loads complete in order among themselves, so "at most N operations outstanding" still implies that every load
//     older than the N youngest loads has landed, whatever the stores do).  The drain overlaps MFMA work, and since no
//     workgroup ever waits at a launch boundary again the CUs fall out of lockstep and the write bursts disappear.
//   * Edge tiles are SHIFTED inside the matrix (m0 = min(tm*256, M-256), same for n): every tile is an interior tile for
//     loads, the K loop and the epilogue (no clamped 64-bit addressing, no predicates).  The overlap region is computed by
//     two tiles with identical fp32 accumulation order, hence written twice with identical bits.
//   * The epilogue is staged through a separate 32 KB of LDS (4 KB per wave, four passes) because the ring is live.
//   * HALF TILES (round 6; a kernel of its own, gemm_nt_4phase_persist_half_kernel, launched when N % 256 == 128 with a bf16 epilogue:
//     the predictor's N = 384 and 1152).  The shifted last column tile would recompute 128 columns its neighbour owns.  Such a tile
//     computes its last 128 columns only, and its four live wave tiles (2 x 2 of 128 x 64) are taken by the EARLY wave group, one per
//     SIMD, while the late group only issues its share of the LDS-DMA and meets the barriers: the intervals in which the late group
//     would compute shrink to the early group's fragment loads, and the epilogue runs with one wave per SIMD.  Every output element
//     still sees the same MFMA sequence over K, so results stay bit-identical; the overlap columns are written once instead of twice.
//     A half tile costs ~0.8 of a full one (its load intervals are bound by the LDS bandwidth of the live waves' fragment reads):
//     N = 384 launches -8 ... -12 %, the step -0.5 ms (profiles/r06_gemm_half_tiles.md).  Option gemm_persist = 3 is the A/B control.
//
// Section structure of one tile (h = ring half of its first K-tile; parts of K-tile t: B0, B1, A1 in half h(t) slots 0..2,
// A0(t+1) in half h(t) slot 3):
//   start    issue part 6 [A1(1)], barrier(s), L(-1): ra0 <- A0(0)
//   HEAD0    K-tile 0: nothing to issue except A0(2) in phase 3; no waits (parts -1..5 landed before the epilogue's vmcnt(0))
//   HEAD1    K-tile 1: steady issues; phase 0 needs no wait (part 5), from phase 1 on the steady vmcnt(6)
//   STEADY   K-tiles 2 .. nk-3
//   TAIL0    K-tile nk-2: phase 3 issues the NEXT tile's A0(0)
//   TAIL1    K-tile nk-1: phases 0..3 issue the next tile's B0(0) | B1(0) | A1(0) + B0(1) | A0(1) + B1(1); the two extra
//            parts go to slots of THIS K-tile whose fragments were read two sections earlier (WAR rule of the template)
//   epilogue bias loads + vmcnt(0) (every DMA issued so far has landed for this wave), convert, stage, store
// Requirements (checked by the launcher, otherwise the one-tile kernel runs): bf16 output epilogues, M, N >= 256,
// K % 64 == 0, K >= 256, more tiles than CUs, 16-byte-aligned rows of C (and aux), C not aliasing the residual.
#include <type_traits>
#include <atomic>
#include "gemm_common.hpp"
#include "options.hpp"

#define PP_PART 16384
#define PP_RING (8 * PP_PART)
#define PP_STAGE_PER_WAVE 4096

namespace {

__device__ __forceinline__ void pp_bar() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);   // nothing (MFMA, ds_read, DMA issue) may be scheduled across a section boundary
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void pp_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ unsigned pp_lds(const char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
// LDS-DMA, SGPR base + 32-bit lane offset form, through inline assembly (the compiler neither counts it nor assumes an LDS write).
// (a non-temporal hint on the streaming operand was measured in round 5: +1.4 / +2.8 ms per step, profiles/r05_gemm_nt.md)
__device__ __forceinline__ void pp_dma(const char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

// the 8 wave-uniform row-group bases of one tile: A rows m0 + j*128 + mq*64, B rows n0 + j*128 + nq*32 (byte pointers at k = 0)
struct PpBases {
  const char* a[2][2];   // [mq][j]
  const char* b[2][2];   // [nq][j]
};
__device__ __forceinline__ void pp_make_bases(PpBases& pb, const char* a0, const char* b0, int64_t lda2, int64_t ldb2) {
#pragma unroll
  for (int sub = 0; sub < 2; sub++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      pb.a[sub][j] = a0 + (int64_t)(j * 128 + sub * 64) * lda2;
      pb.b[sub][j] = b0 + (int64_t)(j * 128 + sub * 32) * ldb2;
    }
}

// logical tile index -> shifted tile origin (always a full 256 x 256 tile inside the matrix)
__device__ __forceinline__ void pp_tile_origin(const GemmArgs& p, int logical, int64_t& m0, int64_t& n0, int& tm) {
  int tn;
  tile_of_raster(logical, p.tiles_m, p.tiles_n, p.raster, tm, tn);
  m0 = (int64_t)tm * 256;
  n0 = (int64_t)tn * 256;
  m0 = m0 + 256 <= p.M ? m0 : p.M - 256;
  n0 = n0 + 256 <= p.N ? n0 : p.N - 256;
}

enum { PP_STEADY = 0, PP_HEAD0, PP_HEAD1, PP_TAIL0, PP_TAIL1 };

// One K-tile = TWO {load section, barrier, compute section, barrier} pairs of 32 MFMAs (k_tile4 below; the four-pair schedule of round 3
// is bit-identical and 0.3 ms per step slower: profiles/r04_abab_gemm_sched4.md).
// PRE: form of the epilogue (gemm_common.hpp gemm_epilogue_staged; option gemm_epi_pre: 4 = pipelined passes, 0 = straight passes)
// STAMP (diagnostics, gemm_dbg bit 2, tools/gemm_stamps.py): waves 0 and 4 (one per wave group) time the phases of every tile with
// s_memtime -- tile start, K loop, wait for the cross-tile prefetch, epilogue, first barrier of the next tile -- and leave the sums in
// the first bytes of C when the workgroup ends.  A kernel of its own: the product kernels carry no trace of it.
template <int EPI, int PRE = 4, bool STAMP = false, bool HALF = false>
__device__ __forceinline__ void pp_body(const GemmArgs& p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 2, wn = wave_u & 3;
  const bool late_group = wave_u >= 4;        // waves 4-7 run one barrier interval behind waves 0-3
  const bool stamper = STAMP && (wave_u & 3) == 0;   // wave-uniform
  unsigned long long st_sum[5] = {0, 0, 0, 0, 0}, st_prev = 0;
  unsigned st_tiles = 0;
  auto lap = [&](int k, bool count) __attribute__((always_inline)) {
    if constexpr (STAMP) {
      if (stamper) {
        const unsigned long long now = __builtin_readcyclecounter();
        if (count) st_sum[k] += now - st_prev;
        st_prev = now;
      }
    }
  };
  const int frow = lane & 15, fg = lane >> 4;

  // ---- this workgroup's tile list: XCD x (= blockIdx & 7, the hardware's round-robin) owns a contiguous band of the
  //      grouped tile order; its workgroups take the band's tiles round-robin, so the tiles an XCD works on at any time
  //      are neighbours (8 row tiles x 4 column tiles: shared A / B panels in its L2)
  const int ntile = p.tiles_m * p.tiles_n;
  const int G = (int)gridDim.x, bid = (int)blockIdx.x;
  const int xcd = bid & 7, pos = bid >> 3;
  const int qx = ntile >> 3, rx = ntile & 7;
  const int band0 = xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx;
  const int band_n = qx + (xcd < rx ? 1 : 0);
  const int wgs_x = (G - xcd + 7) >> 3;               // workgroups of this launch that sit on XCD `xcd`
  if (pos >= band_n) return;                          // (only when there are fewer tiles than workgroups)

  const int nk = (int)(p.K / 64);
  const int64_t lda2 = p.lda * 2, ldb2 = p.ldb * 2;
  // per-thread byte offsets of a part's chunks (see gemm8.hip): row r0 (+ 64 j), chunk column swizzled by the row
  unsigned voff_a, voff_b;
  {
    const int r0 = tid >> 3, cpos = tid & 7;
    const int c = cpos ^ (r0 & 7);
    const int rb = (r0 >> 5) * 64 + (r0 & 31);        // B parts interleave the four wave columns' 32-row halves
    voff_a = (unsigned)((r0 * p.lda + c * 8) * 2);
    voff_b = (unsigned)((rb * p.ldb + c * 8) * 2);
  }
  const unsigned ring = pp_lds(smem);
  const unsigned dst_lane = (unsigned)(wave_u * 64) * 16;   // this wave's 1 KB piece inside each 8 KB half-part

  // issue one part (two DMA instructions per thread): kind 0 B0, 1 B1, 2 A1, 3 A0 of K-tile `kt` into ring half `half`
  auto issue = [&](auto kind_tag, const PpBases& bs, int kt, int half) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_tag)::value;
    const unsigned slot = ring + (unsigned)(half * 4 + KIND) * PP_PART + dst_lane;
    const unsigned koff = (unsigned)kt * 128u;
    if constexpr (KIND == 0 || KIND == 1) {
      pp_dma(bs.b[KIND][0], voff_b + koff, slot);
      pp_dma(bs.b[KIND][1], voff_b + koff, slot + 512 * 16);
    } else {
      constexpr int MQ = KIND == 2 ? 1 : 0;
      pp_dma(bs.a[MQ][0], voff_a + koff, slot);
      pp_dma(bs.a[MQ][1], voff_a + koff, slot + 512 * 16);
    }
  };
  // the same for the NEXT tile (tail of the K loop, once per tile): only its two origin pointers are kept in SGPRs, the
  // row-group bases are derived at the issue site (a handful of SALU instructions per part)
  auto issue_next = [&](auto kind_tag, const char* a0n, const char* b0n, int kt, int half) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_tag)::value;
    const unsigned slot = ring + (unsigned)(half * 4 + KIND) * PP_PART + dst_lane;
    const unsigned koff = (unsigned)kt * 128u;
    if constexpr (KIND == 0 || KIND == 1) {
      pp_dma(b0n + (int64_t)(KIND * 32) * ldb2, voff_b + koff, slot);
      pp_dma(b0n + (int64_t)(128 + KIND * 32) * ldb2, voff_b + koff, slot + 512 * 16);
    } else {
      constexpr int MQ = KIND == 2 ? 1 : 0;
      pp_dma(a0n + (int64_t)(MQ * 64) * lda2, voff_a + koff, slot);
      pp_dma(a0n + (int64_t)(128 + MQ * 64) * lda2, voff_a + koff, slot + 512 * 16);
    }
  };
  using K_B0 = std::integral_constant<int, 0>;
  using K_B1 = std::integral_constant<int, 1>;
  using K_A1 = std::integral_constant<int, 2>;
  using K_A0 = std::integral_constant<int, 3>;

  // per-lane fragment byte offsets inside a part (rows are 128 B, chunks XOR-swizzled by row & 7), set per tile by set_role()
  int a_off[4][2], b_off[2][2];
  // role of this wave in the CURRENT tile: (wm, wn) in a full tile; in a half tile the early group's waves 0..3 take the wave tiles
  // (0,2) (0,3) (1,2) (1,3) and the late group is idle.  The offsets are re-derived from a VOLATILE lane id at every tile start so
  // that they are dead during the epilogue (carried across it they cost the epilogue 12 VGPRs, i.e. spills).
  int wm_t = wm, wn_t = wn;
  bool live = true;
  auto set_role = [&](int64_t n0_tile) __attribute__((always_inline)) {
    const bool half = HALF && ((int)n0_tile & 255) == 128;
    wm_t = (half && !late_group) ? ((wave_u >> 1) & 1) : wm;
    wn_t = (half && !late_group) ? (2 + (wave_u & 1)) : wn;
    live = !(half && late_group);
    int tl = lane;
    if constexpr (HALF) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tl));
    const int fr = tl & 15, fgq = tl >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int c = ks * 4 + fgq;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int pr = wm_t * 64 + i * 16 + fr;
        a_off[i][ks] = pr * 128 + ((c ^ (pr & 7)) * 16);
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int pr = wn_t * 32 + j * 16 + fr;
        b_off[j][ks] = pr * 128 + ((c ^ (pr & 7)) * 16);
      }
    }
  };

  f32x4_t acc[8][4];
  bf16x8_t ra0[4][2], ra1[4][2], rb0[2][2], rb1[2][2];  // [fragment][k-step]

  auto read_a = [&](bf16x8_t (&ra)[4][2], const char* slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) ra[i][ks] = *(const bf16x8_t*)(slot + a_off[i][ks]);
  };
  auto read_b = [&](bf16x8_t (&rb)[2][2], const char* slot) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) rb[j][ks] = *(const bf16x8_t*)(slot + b_off[j][ks]);
  };
  auto mma = [&](auto quad_tag, const bf16x8_t (&rb)[2][2], const bf16x8_t (&ra)[4][2]) __attribute__((always_inline)) {
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

  PpBases cur;                              // wave-uniform (SGPR) row-group base pointers of the current tile
  const char *a0n = nullptr, *b0n = nullptr;   // origin pointers of the next tile

  using M_STEADY = std::integral_constant<int, PP_STEADY>;
  using M_HEAD0 = std::integral_constant<int, PP_HEAD0>;
  using M_HEAD1 = std::integral_constant<int, PP_HEAD1>;
  using M_TAIL0 = std::integral_constant<int, PP_TAIL0>;
  using M_TAIL1 = std::integral_constant<int, PP_TAIL1>;

  // ---- one K-tile = TWO {load section, barrier, compute section, barrier}:
  //   LX(t): rb0 <- B0(t), rb1 <- B1(t);  issue A1(t+1), A0(t+2);  CX(t): quadrants 00, 01 (ra0 x rb0, rb1: 32 MFMAs)
  //   LY(t): ra1 <- A1(t), ra0 <- A0(t+1); issue B0(t+2), B1(t+2); CY(t): quadrants 11, 10 (ra1 x rb1, rb0: 32 MFMAs)
  // Why: a section boundary costs a SIMD a fixed ~80 cycles whatever the section holds (barrier release, the partner's load
  // segment next to the MFMAs, the role switch: MI355X_MICROARCH.md "two waves per SIMD", items 5-7; measured here: 337 cycles
  // per 256-cycle MFMA section).  Halving the boundaries per K-tile halves that cost relative to the matrix time (8 x 337 ->
  // 4 x ~595 cycles per K-tile).  Same part STREAM as the 8-section schedule (..., A1(t+1), A0(t+2), B0(t+2), B1(t+2), ...), two
  // parts per load section and half a K-tile earlier, so the prologue / cross-tile prefetch hand the next tile the same seven
  // parts; same fragment registers (a section only loads registers whose last MFMA is behind the previous barrier); same MFMA
  // order per accumulator, hence bit-identical results.  Ring safety: LX(t) writes half h(t+1) slots 2, 3 = A1(t-1) | A0(t),
  // last read in LY(t-1) (>= 2 sections earlier for both wave groups); LY(t) writes half h(t) slots 0, 1 = B0(t) | B1(t), last
  // read in LX(t).  Waits: a load section leaves at most the 8 youngest DMA instructions in flight (the two parts it issued and
  // the two parts of the section before): the parts the NEXT load section reads were issued before those.  K-tile 0 waits for
  // nothing (its parts and K-tile 1's landed before the epilogue's vmcnt(0)), so the previous tile's stores drain under 64 MFMAs.
  // LIVE = false: the K-tile of an idle wave (late group of a half tile): its share of the LDS-DMA, the waits and the barriers only.
  // (Built and measured level or worse, profiles/r06_gemm_half_tiles.md: the idle group issuing BOTH groups' DMA so that the early group's
  //  load sections hold ds_reads only -- those sections are bound by the LDS bandwidth of the four live waves' fragment reads.)
  auto k_tile4 = [&](auto mode_tag, auto live_tag, int t, int h) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool LIVE = decltype(live_tag)::value;
    constexpr bool HEAD0 = MODE == PP_HEAD0, TAIL0 = MODE == PP_TAIL0, TAIL1 = MODE == PP_TAIL1;
    const char* half_c = smem + h * 4 * PP_PART;
    const int ho = h ^ 1;
    // ---------------- LX
    if constexpr (LIVE) {
      read_b(rb0, half_c + 0 * PP_PART);
      read_b(rb1, half_c + 1 * PP_PART);
    }
    if constexpr (TAIL1) {
      issue_next(K_A1{}, a0n, b0n, 0, ho);
      issue_next(K_A0{}, a0n, b0n, 1, ho);
    } else if constexpr (TAIL0) {
      issue(K_A1{}, cur, t + 1, ho);
      issue_next(K_A0{}, a0n, b0n, 0, ho);
    } else {
      issue(K_A1{}, cur, t + 1, ho);
      issue(K_A0{}, cur, t + 2, ho);
    }
    if constexpr (!HEAD0) pp_wait<8>();
    pp_bar();
    if constexpr (LIVE) {
      mma(Q00{}, rb0, ra0);
      mma(Q01{}, rb1, ra0);
    }
    pp_bar();
    // ---------------- LY
    if constexpr (LIVE) {
      read_a(ra1, half_c + 2 * PP_PART);
      if constexpr (!TAIL1) read_a(ra0, half_c + 3 * PP_PART);   // (the next tile's A0(0) is read at its start, after the epilogue)
    }
    if constexpr (TAIL1) {
      issue_next(K_B0{}, a0n, b0n, 1, h);
      issue_next(K_B1{}, a0n, b0n, 1, h);
    } else if constexpr (TAIL0) {
      issue_next(K_B0{}, a0n, b0n, 0, h);
      issue_next(K_B1{}, a0n, b0n, 0, h);
    } else {
      issue(K_B0{}, cur, t + 2, h);
      issue(K_B1{}, cur, t + 2, h);
    }
    if constexpr (!HEAD0 && !TAIL1) pp_wait<8>();
    pp_bar();
    if constexpr (LIVE) {
      mma(Q11{}, rb1, ra1);
      mma(Q10{}, rb0, ra1);
    }
    pp_bar();
  };

  // ---- first tile: parts -1 .. 5 in flight, all landed before anybody reads
  int64_t m0, n0;
  int tm_cur;                                  // un-shifted row-tile index of the current tile (column-sum slot / row ownership)
  pp_tile_origin(p, band0 + pos, m0, n0, tm_cur);
  pp_make_bases(cur, (const char*)(p.A + m0 * p.lda), (const char*)(p.B + n0 * p.ldb), lda2, ldb2);
  int h = 0;                                   // ring half of the current tile's K-tile 0 (toggles every K-tile, across tiles)
  issue(K_A0{}, cur, 0, 1);                    // part -1: A0(0) -> half h^1, slot 3
  issue(K_B0{}, cur, 0, 0);
  issue(K_B1{}, cur, 0, 0);
  issue(K_A1{}, cur, 0, 0);
  issue(K_A0{}, cur, 1, 0);
  issue(K_B0{}, cur, 1, 1);
  issue(K_B1{}, cur, 1, 1);
  pp_wait<0>();

  // li_cur / li_nx: this workgroup's current and next tile, as positions in its XCD's band (static round-robin lists: li_nx = li_cur +
  // wgs_x; handing tiles out dynamically from per-XCD counters was built in round 5 and measured level: profiles/r05_gemm_dyn.md)
  int li_cur = pos, li_nx = pos + wgs_x;
  using R_LIVE = std::true_type;
  using R_IDLE = std::false_type;
  if constexpr (!HALF) set_role(0);   // (the kernels without half tiles keep their fragment offsets for the whole launch, as before round 6)
  while (true) {
    if constexpr (HALF) set_role(n0);
    bool has_next;                          // workgroup-uniform
    int64_t m0n, n0n;
    int tm_next;
    auto next_origin = [&]() __attribute__((always_inline)) {   // (behind the tile's first barrier, where it has always been)
      has_next = li_nx < band_n;
      pp_tile_origin(p, band0 + (has_next ? li_nx : li_cur), m0n, n0n, tm_next);
      a0n = (const char*)(p.A + m0n * p.lda);
      b0n = (const char*)(p.B + n0n * p.ldb);
    };
    // The diamond is around the WHOLE tile (nothing wide is live across it: accumulators and fragments belong to the live side);
    // branches around single sections made hipcc spill several hundred VGPRs.
    auto compute_tile = [&](auto role_tag) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      // ---- tile start: part 6 = A1(1) -> half h^1 slot 2 (its previous content, A1 of the previous tile's last K-tile, was
      //      read >= 2 sections + one epilogue ago); every other part up to 5 landed before this wave's last vmcnt(0)
      pp_bar();   // (part 6 is issued by LX(0))
      lap(4, st_tiles != 0);       // [epilogue end -> past the next tile's first barrier: the skew between the eight waves]
      next_origin();
      if (late_group) pp_bar();
      read_a(ra0, smem + ((h ^ 1) * 4 + 3) * PP_PART);   // L(-1): A0(0)
      pp_bar();
      pp_bar();
      lap(0, st_tiles != 0);   // [tile start: first barrier -> K loop] (the first tile starts its clock here)
      // ONE tail for every tile: the last tile "prefetches" its own first parts again (never read; retired by the epilogue's
      // vmcnt(0)).  A has_next diamond around two copies of the tail costs ~300 spilled VGPRs (hipcc 7.2).
      k_tile4(M_HEAD0{}, role_tag, 0, h);
      for (int t = 1; t < nk - 2; t++) k_tile4(M_STEADY{}, role_tag, t, h ^ (t & 1));
      k_tile4(M_TAIL0{}, role_tag, nk - 2, h ^ (nk & 1));
      k_tile4(M_TAIL1{}, role_tag, nk - 1, h ^ ((nk - 1) & 1));
      if (!late_group) pp_bar();   // the early group matches the late group's extra barrier
      lap(1, true);                // [K loop]

      // ---- epilogue: staged through this wave's private 4 KB (the ring holds the next tile's parts).  Its first action
      //      (bias loads + s_waitcnt vmcnt(0)) also retires every LDS-DMA this wave has issued.
      if (p.dbg & 1) {   // diagnostics: no output traffic (keeps the accumulators alive through one predicated store)
        pp_wait<0>();
        if (acc[0][0][0] == 12345.678f && acc[7][3][3] == 0.5f) *(float*)p.C = acc[3][2][1];
      } else {
        // lane id re-derived through a VOLATILE asm: everything the epilogue computes from it (LDS staging offsets, row /
        // column addresses) is then re-computed per tile instead of being hoisted out of the tile loop, where it would
        // sit in VGPRs across the K loop (the K loop owns all 256)
        int elane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
        const int efrow = elane & 15, efg = elane >> 4;
        pp_wait<0>();
        lap(2, true);              // [wait for the next tile's prefetched parts]
        // (column sums, EPI_DGELU with p.colpart: a shifted edge tile owns only its rows >= tm * 256; slot = 2 tm + wave row)
        (void)gemm_epilogue_try_staged<EPI, 2, true, PRE>(p, acc, m0 + wm_t * 128, n0 + wn_t * 64, efrow, efg, elane,
                                                          smem + PP_RING + wave_u * PP_STAGE_PER_WAVE, (int64_t)tm_cur * 256, tm_cur * 2 + wm_t);
      }
      lap(3, true);                // [epilogue: bias (+ operand) latency, convert, stage, store issue]
    };
    if (!HALF || live) {
      compute_tile(R_LIVE{});
    } else {
      // idle wave of a half tile (always of the late group): the late group's four tile-start barriers, its DMA share, the waits
      pp_bar();
      next_origin();
      pp_bar();
      pp_bar();
      pp_bar();
      k_tile4(M_HEAD0{}, R_IDLE{}, 0, h);
      for (int t = 1; t < nk - 2; t++) k_tile4(M_STEADY{}, R_IDLE{}, t, h ^ (t & 1));
      k_tile4(M_TAIL0{}, R_IDLE{}, nk - 2, h ^ (nk & 1));
      k_tile4(M_TAIL1{}, R_IDLE{}, nk - 1, h ^ ((nk - 1) & 1));
      pp_wait<0>();
    }
    h ^= (nk & 1);               // ring half of the next tile's K-tile 0
    if constexpr (STAMP) st_tiles++;
    if (!has_next) break;
    li_cur = li_nx;
    li_nx = li_cur + wgs_x;
    m0 = m0n;
    n0 = n0n;
    tm_cur = tm_next;
    pp_make_bases(cur, a0n, b0n, lda2, ldb2);
  }
  if constexpr (STAMP) {
    if (stamper && lane == 0) {   // 64 bytes per (workgroup, wave group) at the start of C (every tile's real output is older than this)
      unsigned long long* d = (unsigned long long*)p.C + ((int)blockIdx.x * 2 + (wave_u >> 2)) * 8;
#pragma unroll
      for (int k = 0; k < 5; k++) d[k] = st_sum[k];
      d[5] = st_tiles;
      d[6] = (unsigned long long)nk;
      d[7] = 0x5354414d50ull;
    }
  }
}

template <int EPI, int PRE>
__global__ __launch_bounds__(512) void gemm_nt_4phase_persist_pre_kernel(GemmArgs p) {
  pp_body<EPI, PRE>(p);
}
template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_4phase_persist_stamp_kernel(GemmArgs p) {
  pp_body<EPI, 4, true>(p);
}
// the kernel with half tiles (N % 256 == 128; bf16 epilogues only: the predictor's proj / fc2 / dgrad / qkv GEMMs)
__global__ __launch_bounds__(512) void gemm_nt_4phase_persist_half_kernel(GemmArgs p) {
  pp_body<EPI_BF16, 4, false, true>(p);
}

int g_num_cus = 0;   // CU count of the (homogeneous) GPUs of this node, read once

}  // namespace

// true when the persistent kernel can run this problem (otherwise the caller keeps the one-tile-per-workgroup kernel)
template <int EPI>
static bool persist_ok(const GemmArgs& a, int ncu) {
  if (EPI == EPI_F32) return false;
  if (a.M < 256 || a.N < 256 || a.K % 64 != 0 || a.K < 256) return false;
  // (single-round shapes, tiles <= CUs, come here too: nothing to overlap across tiles, but this kernel's issue path is
  //  leaner than gemm8.hip's -- SQ_INSTS_SALU / SQ_INSTS_MFMA 0.46 vs 0.96 -- : -0.71 ms/step, profiles/r03_abab_persist_small.md)
  if (a.N % 8 != 0 || a.ldc % 8 != 0 || ((uintptr_t)a.C & 15) != 0) return false;
  if (EPI == EPI_GELU && a.aux_out != nullptr && (a.ldaux % 8 != 0 || ((uintptr_t)a.aux_out & 15) != 0)) return false;
  if (a.lda >= (1 << 23) || a.ldb >= (1 << 23)) return false;            // 32-bit per-lane byte offsets
  // shifted edge tiles recompute (and rewrite, bit-identically) rows / columns of their neighbours: inputs must not alias C
  auto overlaps = [&](const void* q, int64_t ld) {
    if (q == nullptr) return false;
    const char* c0 = (const char*)a.C;
    const char* c1 = c0 + (a.M * a.ldc) * 2;
    const char* q0 = (const char*)q;
    const char* q1 = q0 + (a.M * ld) * 2;
    return q0 < c1 && c0 < q1;
  };
  if (overlaps(a.res, a.ldr) || overlaps(a.aux_in, a.ldaux) || overlaps(a.A, a.lda)) return false;
  if (a.colpart != nullptr && (EPI != EPI_DGELU || ((uintptr_t)a.colpart & 15) != 0)) return false;
  return true;
}

template <int EPI>
static int launch8p(const GemmArgs& a, hipStream_t stream) {
  constexpr int smem = PP_RING + 8 * PP_STAGE_PER_WAVE;   // 160 KB: the whole LDS of a CU
  static VjPerDeviceOnce attr_once;   // the dynamic-LDS limit is a per-device attribute of the function
  attr_once([] {
    (void)hipFuncSetAttribute((const void*)gemm_nt_4phase_persist_pre_kernel<EPI, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)gemm_nt_4phase_persist_pre_kernel<EPI, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)gemm_nt_4phase_persist_stamp_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if constexpr (EPI == EPI_BF16)
      (void)hipFuncSetAttribute((const void*)gemm_nt_4phase_persist_half_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  });
  GemmArgs b = a;
  b.tiles_m = (int)cdiv64(a.M, 256);
  b.tiles_n = (int)cdiv64(a.N, 256);
  b.splitk = 1;
  b.ws = nullptr;
  b.ktiles_per = (int)(a.K / 64);
  b.raster = vj_opt(VJ_OPT_GEMM_RASTER);
  b.epi_pre = vj_opt(VJ_OPT_GEMM_EPI_PRE);
  // half tiles (N % 256 == 128, bf16 epilogues, pipelined form): a kernel of its own, so that every other launch runs the code it ran before
  b.half_tiles = (EPI == EPI_BF16 && a.N % 256 == 128 && b.epi_pre != 0 && vj_opt(VJ_OPT_GEMM_PERSIST) != 3 && !(a.dbg & 4)) ? 1 : 0;
  if (b.raster == 511) {   // automatic: column groups of six for the encoder shapes (K >= 1024), the row-grouped order for the short-K predictor shapes
    b.raster = a.K >= 1024 ? 256 + 6 : 0;
  }
  const int64_t tiles = (int64_t)b.tiles_m * b.tiles_n;
  // Grid: one workgroup per CU (option gemm_persist = 2, the default since late round 6).  Rounds 3 - 6 launched the SMALLEST grid that still needs
  // only `rounds` = ceil(tiles / CUs) tiles per workgroup (588 tiles: 200 workgroups x 3 tiles, 56 CUs left to the step's second stream for the
  // whole launch; option value 1): re-measured at the end of round 6 the full grid is 0.2 ms per step faster in 9 of 10 interleaved rounds
  // (profiles/r06_abab_grid.md) -- the workgroups that run out of tiles a round early leave the last round to fewer CUs at a higher clock.
  // Equal workgroup counts per XCD keep the bands balanced.
  int64_t per_xcd = g_num_cus / 8;
  if (vj_opt(VJ_OPT_GEMM_PERSIST) != 2) {
    const int64_t rounds = cdiv64(tiles, g_num_cus);
    const int64_t band = cdiv64(tiles, 8);                  // tiles of the largest XCD band
    per_xcd = cdiv64(band, rounds);
    if (per_xcd * 8 > g_num_cus) per_xcd = g_num_cus / 8;
  }
  // half tiles cost ~0.8 of a full tile and alternate with full tiles in the order when there are two column tiles (N = 384): a
  // workgroup takes every per_xcd-th tile of its band, so an ODD stride hands everybody both kinds
  if (b.half_tiles && b.tiles_n == 2 && (per_xcd & 1) == 0 && per_xcd > 1) per_xcd += per_xcd * 8 < g_num_cus ? 1 : -1;
  const int grid = (int)(per_xcd * 8);
  if (b.half_tiles) {
    if constexpr (EPI == EPI_BF16) hipLaunchKernelGGL(gemm_nt_4phase_persist_half_kernel, dim3(grid), dim3(512), smem, stream, b);
  } else if (a.dbg & 4) hipLaunchKernelGGL((gemm_nt_4phase_persist_stamp_kernel<EPI>), dim3(grid), dim3(512), smem, stream, b);   // diagnostics: phase stamps
  else if (b.epi_pre == 0) hipLaunchKernelGGL((gemm_nt_4phase_persist_pre_kernel<EPI, 0>), dim3(grid), dim3(512), smem, stream, b);   // A/B control
  else hipLaunchKernelGGL((gemm_nt_4phase_persist_pre_kernel<EPI, 4>), dim3(grid), dim3(512), smem, stream, b);
  VJ_LAUNCH_CHECK("vj_gemm_bf16_nt(persistent 256x256)");
  return 0;
}

// entry used by gemm.hip's dispatcher: returns VJ_PERSIST_NA when the persistent kernel does not apply (the caller falls
// back to the one-tile-per-workgroup kernel), 0 after a launch, a hipError_t when the launch failed
int vj_gemm_launch_8phase_persist(const GemmArgs& a, int epilogue, hipStream_t stream) {
  if (g_num_cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      return -100;
    g_num_cus = n;
  }
  switch (epilogue) {
    case EPI_BF16: return persist_ok<EPI_BF16>(a, g_num_cus) ? launch8p<EPI_BF16>(a, stream) : -100;
    case EPI_GELU: return persist_ok<EPI_GELU>(a, g_num_cus) ? launch8p<EPI_GELU>(a, stream) : -100;
    case EPI_DGELU: return persist_ok<EPI_DGELU>(a, g_num_cus) ? launch8p<EPI_DGELU>(a, stream) : -100;
    default: return -100;
  }
}
