// Weight-gradient GEMM without operand transposes:  dW[N1,N2] (fp32) = alpha * dY[T,N1]^T X[T,N2] + beta * dW
//
// Replaces the "transpose dY, transpose X, then C = A B^T" route for every nn.Linear weight gradient of the V-JEPA step
// (reference: autograd of nn.Linear, modules.py:31-34,63,76; the optimizer reads these fp32 gradients, train.py:461-476).
// Both operands are read exactly as the backward pass produced them -- row = token, columns = features -- so the
// contraction index (the token) is the SLOW dimension of both.  gfx950's ds_read_b64_tr_b16 delivers an MFMA operand
// fragment whose k index runs down the rows of a row-major LDS image, which makes this layout free:
//
//   * same schedule as gemm8.hip (256 x 256 tile, 8 waves, four 16 KB parts per 64-token K-tile in an 8-slot LDS ring,
//     prefetch distance 4 parts, counted vmcnt across raw barriers, two wave groups one barrier interval apart);
//   * a part is 64 tokens x 128 feature columns (256-byte rows).  Part A0/A1 = columns [0,128) / [128,256) of the dY
//     tile, B0/B1 the same of the X tile; wave (wm, wn) owns output rows {wm*64 + [0,64)} of each A half and columns
//     {wn*32 + [0,32)} of each B half, i.e. four 64 x 32 quadrants (acc[ih][jh]);
//   * 16-byte chunk c of token row r sits at chunk c ^ ((r & 7) << 1): the DMA destination is lane-linear, so the
//     swizzle is applied to the per-lane SOURCE column; the transpose read of a lane group (8 token rows x 32 bytes)
//     then touches all 64 banks once;
//   * a fragment = two transpose reads (token rows 4g..4g+3 and 16+4g..16+4g+3 of a 32-token step).  The k-slot ->
//     token map differs from a plain row read but is the same for both operands, which is all a dot product needs;
//   * tokens beyond T (last K-tile) contribute zero: dY rows are redirected to a zero row (X rows re-read row T-1);
//   * the LDS-DMA is issued through inline assembly so that the compiler does not put an `s_waitcnt vmcnt(0)` in front
//     of the transpose reads (it does after the builtin form; see attention.hip).
//
// Split-K over the token tiles and the fp32 epilogue are shared with the NT kernels (gemm_common.hpp).
#include <type_traits>
#include "gemm_common.hpp"
#include "options.hpp"

#define TN_PART_BYTES 16384

namespace {

// 128 bf16 zeros: the LDS-DMA source of token rows beyond T (the last, partial token tile).  A zero-initialised device global:
// it exists on every device the code object is loaded on, needs no allocation, no lock and no synchronisation on first use
// (the round-4 host-allocated buffer took a global mutex per launch and a device-wide synchronise on first use, which would
// have invalidated a stream capture in progress).
__device__ __attribute__((aligned(256))) bf16_t g_tn_zero_row[128];

__device__ __forceinline__ void tn_cfence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void tn_bar() {
  tn_cfence();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  tn_cfence();
}
template <int N>
__device__ __forceinline__ void tn_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ unsigned tn_lds_addr(const char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void tn_dma16(const void* gsrc, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_base), "v"(gsrc) : "memory");
}

typedef __attribute__((ext_vector_type(4))) short tn_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short tn_s16x8_t;

// fragment: 16 feature columns x 32 tokens, from the part image at `p` (= slot + lane offset + token-step offset)
__device__ __forceinline__ bf16x8_t tn_frag(const char* p) {
  const tn_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4_t*)p);
  const tn_s16x4_t hi =
      __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4_t*)(p + 16 * 256));
  const tn_s16x8_t w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, w);
}

struct TnLane {   // per-lane constants of the part image (2 VGPRs live across the K loop)
  int row0;   // token row of this lane's chunk for j = 0 (0..31); j = 1 is 32 rows further, same chunk
  int sc8;    // source column (elements) inside the 128-column half: ((tid & 15) ^ ((row0 & 7) << 1)) * 8
};

// issue the LDS-DMA of part q (q = -1 .. 4*nk-1) into slot (q+8) % 8.  qq = q + 1 = 4T + kind:
// kind 0 = A0(T), 1 = B0(T), 2 = B1(T), 3 = A1(T)  (A = dY tile, B = X tile, T = token tile).
// One branch-free form for every tile: uniform base pointer + 32-bit per-lane offset; token rows beyond T are clamped
// to row T-1 (a valid address) and, for dY, redirected to the zero row so that they contribute nothing.
__device__ __forceinline__ void tn_issue_part(const GemmArgs& p, int q, int64_t m0, int64_t n0, int kt0, char* smem,
                                              const TnLane& tl, int wave_u) {
  const int qq = q + 1;
  const int kind = qq & 3;
  const int t = qq >> 2;
  const int64_t tok0 = (int64_t)(kt0 + t) * 64;
  char* slot = smem + ((q + 8) & 7) * TN_PART_BYTES;
  const bool isA = (kind == 0) || (kind == 3);
  const int half = (kind == 0 || kind == 1) ? 0 : 1;
  const int64_t cbase = (isA ? m0 : n0) + half * 128;
  // columns beyond the matrix re-read its last 8 columns (those outputs are never stored); may be negative
  const int lim = (int)((isA ? p.M : p.N) - cbase - 8);
  const int rel = tl.sc8 < lim ? tl.sc8 : lim;
  const unsigned ld = (unsigned)(isA ? p.lda : p.ldb);
  const bf16_t* base = (isA ? p.A + tok0 * p.lda : p.B + tok0 * p.ldb) + cbase;   // uniform
  const int64_t left = p.K - 1 - tok0;
  const int last_row = left < 63 ? (int)left : 63;   // uniform: last existing token row of this tile
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int row = tl.row0 + 32 * j;
    const int rowc = row < last_row ? row : last_row;
    const bf16_t* src = base + (int64_t)(int)(__umul24((unsigned)rowc, ld) + rel);
    if (isA) {
      const bf16_t* zsrc = g_tn_zero_row + tl.sc8;
      src = row <= last_row ? src : zsrc;
    }
    tn_dma16(src, tn_lds_addr(slot + (j * 512 + wave_u * 64) * 16));
  }
}

// ---- strength-reduced issue path of the K loop -------------------------------------------------------------------
// tn_issue_part recomputes tile, kind, clamps and a 64-bit base per call: ~35 scalar + ~15 vector instructions per part,
// 138 SALU per K-tile per wave (SQ_INSTS_SALU / SQ_INSTS_MFMA = 2.15 against 0.46 in the NT kernel), all of it inside the
// load sections whose length sets the pace of the two wave groups.  Inside the loop every kind (A0, B0, B1, A1) is issued
// exactly once per K-tile, so each keeps a running scalar base (token row 0 of the tile it loads next, column cbase) that
// advances by 64 rows per issue; the per-lane part is one v_min (column clamp), one v_lshl_add (row0 * ld + column) and one
// v_add (row0 + 32).  The generic form stays for the prologue, for a half that lies entirely outside the matrix and for the
// last, partial token tile (row clamp / zero row).
struct TnFast {
  const char* base[4];    // per kind, bytes
  int lim[4];             // per kind: last admissible source column (elements, relative to cbase); < 0: generic path only
  unsigned step[2];       // bytes per K-tile (64 rows): [0] = A (dY), [1] = B (X)
  unsigned row32[2];      // bytes of 32 rows
  int tail_tile;          // slice-relative index of the partial token tile, or -1
};
__device__ __forceinline__ void tn_dma_sv(const char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}
template <int KIND>
__device__ __forceinline__ void tn_issue_fast(TnFast& f, int q, char* smem, int sc8, unsigned rowoff_a, unsigned rowoff_b,
                                              int wave_u) {
  constexpr bool isA = (KIND == 0) || (KIND == 3);
  const unsigned slot = tn_lds_addr(smem + ((q + 8) & 7) * TN_PART_BYTES) + (unsigned)(wave_u * 64) * 16;
  const int rel = sc8 < f.lim[KIND] ? sc8 : f.lim[KIND];
  const unsigned v0 = (isA ? rowoff_a : rowoff_b) + ((unsigned)rel << 1);
  const unsigned v1 = v0 + f.row32[isA ? 0 : 1];
  tn_dma_sv(f.base[KIND], v0, slot);
  tn_dma_sv(f.base[KIND], v1, slot + 512 * 16);
  f.base[KIND] += f.step[isA ? 0 : 1];
}

// One launch for up to TN_GROUP_MAX problems that share the token count (the four weight gradients of a transformer
// block): work units = (problem, K slice, tile), numbered problem-major; unit_end[i] = cumulative count.
#define TN_GROUP_MAX 4
struct TnGroupArgs {
  int n;
  int unit_end[TN_GROUP_MAX];
  GemmArgs g[TN_GROUP_MAX];
};
struct TnSingleArgs {
  GemmArgs g;
};

template <bool GROUPED>
__global__ __launch_bounds__(512) void gemm_tn_8phase_kernel(std::conditional_t<GROUPED, TnGroupArgs, TnSingleArgs> ga) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 2, wn = wave_u & 3;
  const bool late_group = wave_u >= 4;

  GemmArgs p;
  p.colpart = nullptr;
  p.lnf_rs = nullptr;
  p.lnf_c = nullptr;
  p.raster = 0;
  p.qscale = 0.f;
  p.qcols = 0;
  int logical_all;
  if constexpr (GROUPED) {
    const int l = xcd_logical(blockIdx.x, ga.unit_end[ga.n - 1]);
    int idx = 0;
#pragma unroll
    for (int i = 0; i + 1 < TN_GROUP_MAX; i++)
      if (i + 1 < ga.n && l >= ga.unit_end[i]) idx = i + 1;
    logical_all = l - (idx > 0 ? ga.unit_end[idx - 1] : 0);
    p = ga.g[idx];
  } else {
    p = ga.g;
    logical_all = xcd_logical(blockIdx.x, p.tiles_m * p.tiles_n * p.splitk);
  }
  const int ntile = p.tiles_m * p.tiles_n;
  const int slice = logical_all / ntile;
  int tm, tn;
  tile_of(logical_all - slice * ntile, p.tiles_m, p.tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
  const int nk_all = (int)((p.K + 63) / 64);
  const int kt0 = slice * p.ktiles_per;
  const int nk = (kt0 + p.ktiles_per < nk_all ? kt0 + p.ktiles_per : nk_all) - kt0;
  const int last_part = 4 * nk - 2;

  f32x4_t acc[2][2][4][2];   // [row half][column half][16-row block][16-column block]
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[a][b][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t ra0[4][2], ra1[4][2], rb0[2][2], rb1[2][2];   // [fragment][32-token step]

  // transpose-read source of this lane: token row 4g + jj of a 16-row group, 8-byte piece q4 of the 32-byte column run
  int a_off[4], b_off[2];
  {
    const int g = lane >> 4, jj = (lane & 15) >> 2, q4 = lane & 3;
    const int row_l = 4 * g + jj;
    const int key2 = (row_l & 7) << 1;
    const int base = row_l * 256 + (q4 & 1) * 8;
#pragma unroll
    for (int i = 0; i < 4; i++) a_off[i] = base + ((((wm * 8 + i * 2) ^ key2) | (q4 >> 1)) << 4);
#pragma unroll
    for (int j = 0; j < 2; j++) b_off[j] = base + ((((wn * 4 + j * 2) ^ key2) | (q4 >> 1)) << 4);
  }

  TnLane tl;
  tl.row0 = tid >> 4;
  tl.sc8 = ((tid & 15) ^ (((tid >> 4) & 7) << 1)) * 8;

  // Schedule (round 3).  The transpose reads are two 8-byte LDS instructions per fragment -- 16 or 32 per part -- and with
  // one wave per SIMD issuing them a load section took longer than the 16 MFMAs of the partner group's compute section
  // (744 vs 1028 TF/s for the NT kernel on the same shape).  Now the fragments of part q+1 are read INSIDE compute
  // section C(q), two reads behind every one or two MFMAs (an MFMA 16x16x32 occupies the matrix pipe for 16 cycles, the
  // reads issue underneath), and a load section only issues the LDS-DMA of part q+5 and waits.  Only the B0 fragments
  // (register set rb0, still in use in C(4t+3)) are read in a load section, L(4t+4).
  //   L(q): [q = 4t: rb0 <- part q]  issue part q+5;  s_waitcnt until part q+2 has landed (q+3..q+5 in flight);  barrier
  //   C(q): 16 MFMA (quadrant q & 3)  ||  reads of part q+1 (q & 3 = 0: rb1, 1: ra1, 2: ra0 of the next K-tile, 3: none); barrier
  // Visibility: the early group reads part q+1 in C(q), while the late group is still in L(q) -- so a part must be complete
  // one section earlier than in gemm8.hip: L(j) waits for part j+2, and the late group's L(q-1) (which precedes the barrier
  // that opens the early group's C(q)) has waited for part q+1.  Prefetch distance 5 keeps three parts in flight as before.
  // Slot reuse: part p+8 is issued in L(p+3), at least two barriers after the last read of part p by either group.
  auto wait_landed = [&](int q) {   // tail-safe form of "part q+2 has landed": parts q+3 .. min(q+5, last_part) may be in flight
    const int younger = last_part - (q + 2);
    if (younger >= 3) tn_wait_vm<6>();
    else if (younger == 2) tn_wait_vm<4>();
    else if (younger == 1) tn_wait_vm<2>();
    else tn_wait_vm<0>();
  };
  // 16 MFMAs of one accumulator quadrant with NF fragment loads (two transpose reads each) spread underneath them
  auto mma_rd = [&](f32x4_t (&acc4)[4][2], const bf16x8_t (&rb)[2][2], const bf16x8_t (&ra)[4][2], auto nf_tag, auto&& load_frag)
                    __attribute__((always_inline)) {
    constexpr int NF = decltype(nf_tag)::value;   // 0, 4 (a B set) or 8 (an A set)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[j][ks], ra[i][ks], acc4[i][j], 0, 0, 0);
          constexpr int dummy = 0;
          (void)dummy;
          const int m = ks * 8 + i * 2 + j;
          if constexpr (NF == 8) {
            if (m % 2 == 0) load_frag(m / 2);
          } else if constexpr (NF == 4) {
            if (m % 4 == 0) load_frag(m / 4);
          }
        }
    __builtin_amdgcn_s_setprio(0);
    // pin the interleave: (MFMA x 16/NF, DS read x 2) per fragment -- without it the scheduler hoists all reads to the top
    if constexpr (NF == 8) {
#pragma unroll
      for (int f = 0; f < 8; f++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    } else if constexpr (NF == 4) {
#pragma unroll
      for (int f = 0; f < 4; f++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    }
  };
  using NF0 = std::integral_constant<int, 0>;
  using NF4 = std::integral_constant<int, 4>;
  using NF8 = std::integral_constant<int, 8>;

  // running bases of the strength-reduced issue path: after the prologue (parts -1 .. 4) the next part of kind A0 / B0
  // belongs to K-tile 2 of this slice, of kind B1 / A1 to K-tile 1
  TnFast fs;
  {
    const unsigned sa = (unsigned)p.lda * 128u, sb = (unsigned)p.ldb * 128u;   // 64 rows in bytes
    fs.step[0] = sa;
    fs.step[1] = sb;
    fs.row32[0] = sa >> 1;
    fs.row32[1] = sb >> 1;
#pragma unroll
    for (int kind = 0; kind < 4; kind++) {
      const bool isA = (kind == 0) || (kind == 3);
      const int half = (kind == 0 || kind == 1) ? 0 : 1;
      const int64_t cbase = (isA ? m0 : n0) + half * 128;
      const int first = (kind == 0 || kind == 1) ? 2 : 1;
      const int64_t tok0 = (int64_t)(kt0 + first) * 64;
      fs.base[kind] = (const char*)((isA ? p.A + tok0 * p.lda : p.B + tok0 * p.ldb) + cbase);
      fs.lim[kind] = (int)((isA ? p.M : p.N) - cbase - 8);
    }
    fs.tail_tile = (p.K & 63) ? nk_all - 1 - kt0 : -1;
  }
  const unsigned rowoff_a = (unsigned)tl.row0 * (unsigned)p.lda * 2u, rowoff_b = (unsigned)tl.row0 * (unsigned)p.ldb * 2u;
  auto issue_loop = [&](auto kind_tag, int q, int tile) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_tag)::value;
    if (tile != fs.tail_tile && fs.lim[KIND] >= 0) tn_issue_fast<KIND>(fs, q, smem, tl.sc8, rowoff_a, rowoff_b, wave_u);
    else tn_issue_part(p, q, m0, n0, kt0, smem, tl, wave_u);
  };

  // ---- prologue: parts -1 .. 4 in flight; parts -1, 0, 1 landed for everyone
  tn_issue_part(p, -1, m0, n0, kt0, smem, tl, wave_u);
#pragma unroll
  for (int q = 0; q < 5; q++)
    if (q <= last_part) tn_issue_part(p, q, m0, n0, kt0, smem, tl, wave_u);
  wait_landed(-1);
  tn_bar();
  if (late_group) tn_bar();
  {
    const char* slot = smem + 7 * TN_PART_BYTES;   // part -1: A0 of K-tile 0
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) ra0[i][ks] = tn_frag(slot + a_off[i] + ks * 32 * 256);
  }
  tn_bar();
  tn_bar();

  for (int t = 0; t < nk; t++) {
#pragma unroll
    for (int ph = 0; ph < 4; ph++) {
      const int q = 4 * t + ph;
      // ---------------- L(q)
      if (ph == 0) {
        const char* slot = smem + (q & 7) * TN_PART_BYTES;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int ks = 0; ks < 2; ks++) rb0[j][ks] = tn_frag(slot + b_off[j] + ks * 32 * 256);
      }
      if (q + 5 <= last_part) {
        // part q+5 = kind (ph + 2) & 3 of K-tile t+1 (ph 0, 1) / t+2 (ph 2, 3)
        if (ph == 0) issue_loop(std::integral_constant<int, 2>{}, q + 5, t + 1);
        else if (ph == 1) issue_loop(std::integral_constant<int, 3>{}, q + 5, t + 1);
        else if (ph == 2) issue_loop(std::integral_constant<int, 0>{}, q + 5, t + 2);
        else issue_loop(std::integral_constant<int, 1>{}, q + 5, t + 2);
        tn_wait_vm<6>();                              // part q+2 landed; q+3..q+5 in flight
      } else {
        wait_landed(q);
      }
      tn_bar();
      // ---------------- C(q): one quadrant x 64 tokens, with the reads of part q+1 underneath
      const char* nslot = smem + ((q + 1) & 7) * TN_PART_BYTES;
      if (ph == 0) {
        mma_rd(acc[0][0], rb0, ra0, NF4{}, [&](int f) { rb1[f >> 1][f & 1] = tn_frag(nslot + b_off[f >> 1] + (f & 1) * 32 * 256); });
      } else if (ph == 1) {
        mma_rd(acc[0][1], rb1, ra0, NF8{}, [&](int f) { ra1[f >> 1][f & 1] = tn_frag(nslot + a_off[f >> 1] + (f & 1) * 32 * 256); });
      } else if (ph == 2) {
        // (in the last K-tile this reads a slot whose part does not exist: stale LDS bytes into registers nobody uses --
        //  cheaper than a second copy of the MFMA block behind a branch, which costs ~30 spilled VGPRs)
        mma_rd(acc[1][1], rb1, ra1, NF8{}, [&](int f) { ra0[f >> 1][f & 1] = tn_frag(nslot + a_off[f >> 1] + (f & 1) * 32 * 256); });
      } else {
        mma_rd(acc[1][0], rb0, ra1, NF0{}, [&](int) {});
      }
      tn_bar();
    }
  }
  if (!late_group) tn_bar();

  const int elane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int efrow = elane & 15, efg = elane >> 4;
#pragma unroll
  for (int ih = 0; ih < 2; ih++)
#pragma unroll
    for (int jh = 0; jh < 2; jh++)
      gemm_epilogue<EPI_F32, 4, 2, true>(p, acc[ih][jh], m0 + ih * 128 + wm * 64, n0 + jh * 128 + wn * 32, efrow, efg,
                                         slice);
}

}  // namespace

__global__ void splitk_reduce_kernel(const float4* ws, float* out, int64_t M, int64_t N, int64_t ldc, int S,
                                     float alpha, float beta);   // gemm.hip

static int tn_check(const char* who, const void* dY, int64_t ldy, const void* X, int64_t ldx, const float* dW, int64_t ldw,
                    int64_t T, int64_t N1, int64_t N2) {
  VJ_CHECK_ARG(T > 0, "%s: T must be positive", who);
  VJ_CHECK_ARG(N1 % 8 == 0 && N2 % 8 == 0, "%s: N1=%ld, N2=%ld must be multiples of 8", who, (long)N1, (long)N2);
  VJ_CHECK_ARG(ldy % 8 == 0 && ldx % 8 == 0 && ldy >= N1 && ldx >= N2 && ldy < (1 << 24) && ldx < (1 << 24),
               "%s: ldy/ldx must be multiples of 8, >= N1/N2 and < 2^24", who);
  VJ_CHECK_ARG(((uintptr_t)dY % 16 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)dW % 16 == 0),
               "%s: operands must be 16-byte aligned", who);
  VJ_CHECK_ARG(ldw % 4 == 0 && ldw >= N2, "%s: ldw=%ld must be a multiple of 4 and >= N2", who, (long)ldw);
  return 0;
}

static GemmArgs tn_args(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* dW, int64_t ldw, int64_t T,
                        int64_t N1, int64_t N2, float alpha, float beta) {
  GemmArgs b;
  b.colpart = nullptr;
  b.lnf_rs = nullptr;
  b.lnf_c = nullptr;
  b.raster = 0;
  b.qscale = 0.f;
  b.qcols = 0;
  b.A = (const bf16_t*)dY; b.B = (const bf16_t*)X; b.C = dW; b.bias = nullptr; b.res = nullptr; b.aux_in = nullptr;
  b.aux_out = nullptr;
  b.M = N1; b.N = N2; b.K = T; b.lda = ldy; b.ldb = ldx; b.ldc = ldw; b.ldr = 0; b.ldaux = 0;
  b.alpha = alpha; b.beta = beta;
  b.tiles_m = (int)cdiv64(N1, 256);
  b.tiles_n = (int)cdiv64(N2, 256);
  b.dbg = 0;
  b.zero_row = nullptr;   // (the TN kernel reads the zero-initialised device global g_tn_zero_row)
  b.splitk = 1;
  b.ktiles_per = (int)cdiv64(T, 64);
  b.ws = nullptr;
  return b;
}

static void tn_reduce(const GemmArgs& b, float alpha, float beta, hipStream_t stream) {
  const int64_t n4 = b.M * b.N / 4;
  int64_t g = cdiv64(n4, 256);
  if (g > 256 * 8) g = 256 * 8;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const float4*)b.ws, (float*)b.C, b.M, b.N,
                     b.ldc, b.splitk, alpha, beta);
}

template <bool GROUPED>
static void tn_set_attr() {
  static VjPerDeviceOnce attr_once;   // the dynamic-LDS limit is a per-device attribute of the function
  attr_once([] {
    (void)hipFuncSetAttribute((const void*)gemm_tn_8phase_kernel<GROUPED>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              8 * TN_PART_BYTES);
  });
}

// dW[N1,N2] = alpha * dY[T,N1]^T X[T,N2] + beta * dW.  (backward of every nn.Linear on the path)
extern "C" int vj_gemm_bf16_tn_splitk(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* dW, int64_t ldw,
                                      int64_t T, int64_t N1, int64_t N2, float alpha, float beta, void* ws,
                                      int64_t ws_bytes, hipStream_t stream) {
  VJ_CHECK_ARG(T >= 0 && N1 >= 0 && N2 >= 0, "vj_gemm_bf16_tn_splitk: negative dim");
  if (N1 == 0 || N2 == 0) return 0;
  if (int rc = tn_check("vj_gemm_bf16_tn_splitk", dY, ldy, X, ldx, dW, ldw, T, N1, N2)) return rc;
  VJ_CHECK_ARG(ws != nullptr && ws_bytes >= N1 * N2 * 4, "vj_gemm_bf16_tn_splitk: workspace must hold at least N1*N2 fp32");
  tn_set_attr<false>();
  TnSingleArgs a;
  GemmArgs& b = a.g;
  b = tn_args(dY, ldy, X, ldx, dW, ldw, T, N1, N2, alpha, beta);
  const int nk = (int)cdiv64(T, 64);
  b.splitk = pick_splitk((int64_t)b.tiles_m * b.tiles_n, nk, 256, 1.45, 8, N1, N2, ws_bytes);
  b.ws = (float*)ws;
  b.ktiles_per = (nk + b.splitk - 1) / b.splitk;
  b.splitk = (nk + b.ktiles_per - 1) / b.ktiles_per;
  hipLaunchKernelGGL(gemm_tn_8phase_kernel<false>, dim3(b.tiles_m * b.tiles_n * b.splitk), dim3(512), 8 * TN_PART_BYTES, stream, a);
  VJ_LAUNCH_CHECK("vj_gemm_bf16_tn_splitk");
  if (b.splitk > 1) {
    tn_reduce(b, alpha, beta, stream);
    VJ_LAUNCH_CHECK("vj_gemm_bf16_tn_splitk(reduce)");
  }
  return 0;
}

// The weight gradients of n <= 4 Linear layers over the SAME T tokens in one launch (a transformer block's qkv, proj,
// fc1, fc2: reference = the four nn.Linear backward nodes of Block.forward, modules.py:31-34,63,76).  Four separate
// launches each pay their own pipeline fill, fp32 slice partials, slice reduction and tail; together the block's 192
// (ViT-L) output tiles fill the GPU WITHOUT split-K, and the predictor's 38 tiles share one split factor.  The K-slice
// length is common to the group; problem i's partials live at ws + sum_{j<i} splitk * N1_j * N2_j * 4.
struct vj_tn_problem_abi {
  const void* dY; int64_t ldy;
  const void* X; int64_t ldx;
  float* dW; int64_t ldw;
  int64_t N1, N2;
};
extern "C" int vj_gemm_bf16_tn_grouped(const void* probs_v, int64_t n, int64_t T, float alpha, float beta, void* ws,
                                       int64_t ws_bytes, hipStream_t stream) {
  const vj_tn_problem_abi* pr = (const vj_tn_problem_abi*)probs_v;
  VJ_CHECK_ARG(n >= 0 && n <= TN_GROUP_MAX, "vj_gemm_bf16_tn_grouped: n=%ld (at most %d problems)", (long)n, TN_GROUP_MAX);
  VJ_CHECK_ARG(T >= 0, "vj_gemm_bf16_tn_grouped: negative T");
  TnGroupArgs a;
  a.n = 0;
  int64_t tiles = 0, out_elems = 0;
  for (int64_t i = 0; i < n; i++) {
    VJ_CHECK_ARG(pr[i].N1 >= 0 && pr[i].N2 >= 0, "vj_gemm_bf16_tn_grouped: negative dim");
    if (pr[i].N1 == 0 || pr[i].N2 == 0) continue;
    if (int rc = tn_check("vj_gemm_bf16_tn_grouped", pr[i].dY, pr[i].ldy, pr[i].X, pr[i].ldx, pr[i].dW, pr[i].ldw, T, pr[i].N1,
                          pr[i].N2))
      return rc;
    a.g[a.n] = tn_args(pr[i].dY, pr[i].ldy, pr[i].X, pr[i].ldx, pr[i].dW, pr[i].ldw, T, pr[i].N1, pr[i].N2, alpha, beta);
    tiles += (int64_t)a.g[a.n].tiles_m * a.g[a.n].tiles_n;
    out_elems += pr[i].N1 * pr[i].N2;
    a.n++;
  }
  if (a.n == 0) return 0;
  VJ_CHECK_ARG(ws != nullptr && ws_bytes >= out_elems * 4, "vj_gemm_bf16_tn_grouped: workspace must hold at least sum N1*N2 fp32");
  const int nk = (int)cdiv64(T, 64);
  // one split factor for the group: pick_splitk's cost model on the summed tile count / output size
  int splitk = pick_splitk(tiles, nk, 256, 1.45, 8, out_elems, 1, ws_bytes);
  const int ktiles_per = (nk + splitk - 1) / splitk;
  splitk = (nk + ktiles_per - 1) / ktiles_per;
  int64_t units = 0, ws_off = 0;
  for (int i = 0; i < a.n; i++) {
    a.g[i].splitk = splitk;
    a.g[i].ktiles_per = ktiles_per;
    a.g[i].ws = (float*)((char*)ws + ws_off);
    ws_off += (int64_t)splitk * a.g[i].M * a.g[i].N * 4;
    units += (int64_t)a.g[i].tiles_m * a.g[i].tiles_n * splitk;
    a.unit_end[i] = (int)units;
  }
  for (int i = a.n; i < TN_GROUP_MAX; i++) {
    a.unit_end[i] = (int)units;
    a.g[i] = a.g[a.n - 1];
  }
  VJ_CHECK_ARG(units < (1ll << 31), "vj_gemm_bf16_tn_grouped: grid too large");
  tn_set_attr<true>();
  hipLaunchKernelGGL(gemm_tn_8phase_kernel<true>, dim3((unsigned)units), dim3(512), 8 * TN_PART_BYTES, stream, a);
  VJ_LAUNCH_CHECK("vj_gemm_bf16_tn_grouped");
  if (splitk > 1) {
    for (int i = 0; i < a.n; i++) tn_reduce(a.g[i], alpha, beta, stream);
    VJ_LAUNCH_CHECK("vj_gemm_bf16_tn_grouped(reduce)");
  }
  return 0;
}
