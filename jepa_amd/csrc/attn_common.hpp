// Shared pieces of the attention kernels (attention.hip: forward + the two-kernel backward; attention_bwd_fused.hip: the one-pass
// backward of round 6): LDS-DMA helpers, the swizzled row-major tile image, transposed fragment reads, segment tables.
#pragma once
#include "common.hpp"
#include "options.hpp"
#include "../../include/vjepa_hip.h"
#include <type_traits>
#include <cstdlib>

typedef float f32x2_t __attribute__((ext_vector_type(2)));


__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  bf2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

// 16 bytes per lane, global -> LDS, asynchronous (vmcnt).  Issued through inline assembly on purpose: the compiler
// then does not know that LDS is written behind its back and inserts no conservative `s_waitcnt vmcnt(0)` in front of
// later LDS reads (it does so for ds_read_b64_tr_b16 after the builtin form, which would collapse the prefetch
// distance); completion is tracked by hand with wait_vmcnt<N>() + raw_barrier().  lds_base must be wave-uniform: the
// hardware adds lane * 16.
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_base), "v"(gsrc) : "memory");
}
// the same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: no 64-bit vector address
// arithmetic per instruction (the attention kernels are bound by VALU issue)
__device__ __forceinline__ void dma16_sv(const void* sbase, unsigned voff, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base), "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void raw_barrier() {   // s_barrier without the vmcnt(0) drain of __syncthreads()
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only: this wave's LDS reads of the previous tile are complete
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#define DEFER_LOG2 5.0f  // forward softmax: rescale O only when a row max grows by more than 2^5

// ---- round 4: "seeded" soft-max (template parameter SM = 1; SM = 0 keeps the round-3 arithmetic for A/B) ----------------
// The attention kernels are bound by vector-instruction issue and the matrix and vector pipes of a SIMD add
// (profiles/r03_valu_mfma_probe.md), so only REMOVING vector instructions per score helps.  Per score the round-3 kernels
// spend: 1/2 v_pk_fma (s*sc - m), 1 v_exp, 1/2 v_max3 (forward), 1/2 v_pk_add (row sum), 1/2 v_cvt_pk.  With SM = 1:
//   * the soft-max scale scale*log2(e) is folded into the STATIONARY operand of the score product (Q in the forward and in
//     dQ, K in dK/dV), once per workgroup: bf16(c * x) -- one more bf16 rounding of that operand;
//   * the score accumulators START at -m_run (forward) / -lse2[q] (backward), so the MFMA delivers s - m and v_exp_f32
//     reads the accumulator directly: no per-score FMA;
//   * the forward does not compute a row maximum per tile.  The running base m_run is kept 2^SM_HEADROOM above the largest
//     score seen when it was last set, so every probability is normally <= 2^-SM_HEADROOM; a tile needs a new base only if
//     some probability reaches 2.0, i.e. bit 14 (the top exponent bit) of a packed bf16 P word is set: the test is the OR of
//     the eight packed words of a row against 0x40004000 (v_or3_b32: 5 plain instructions per 16 scores instead of 8 v_max3
//     + two cross-lane exchanges).  OR >= max for unsigned integers, so "bit clear" PROVES every P < 2 (inf / NaN have the
//     bit set and take the slow path, which computes the exact maximum and re-bases exactly like the round-3 code);
//   * head_dim 24 (the predictor; 32-wide class): column 24 of the V image holds 1.0, so the P.V MFMA accumulates the row
//     sum of the bf16-rounded P in output column 24 -- the row sums leave the vector pipe too (template parameter PSUM).
#define SM_HEADROOM 5.0f
__device__ __forceinline__ bf16x8_t scale_frag(bf16x8_t f, float c) {
  u32x4_t w = __builtin_bit_cast(u32x4_t, f);
#pragma unroll
  for (int j = 0; j < 4; j++) w[j] = cvt_pk_bf16(bf_lo(w[j]) * c, bf_hi(w[j]) * c);
  return __builtin_bit_cast(bf16x8_t, w);
}

// 16-byte-chunk XOR key of a row.  128/256-byte rows (hd 64/128): row & 7.  64- and 192-byte rows (hd 32 / 96): rows r
// and r+4 start on the same banks, so the key must separate the four row quads that one ds_read_b128 lane group
// ({0-3,12-15} of one g with {4-11} of the next) or one ds_read_b64_tr_b16 half (rows 0-7) touches: quads 0,1,2,3 get
// keys 0,3,2,1 (only the low two chunk bits flip, so a 12-chunk row stays inside itself).  Unswizzled, the hd<=32
// kernels spent 33-43 % of their LDS cycles in bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
template <int HDP>
__device__ __forceinline__ int rm_swz(int row) {
  if constexpr (HDP % 64 == 0) return row & 7;
  else return (4 - ((row >> 2) & 3)) & 3;
}
// 16-column output tiles of the head dimension: the 96-wide class serves hd <= 80 (ViT-H) with five, not six
template <int HDP>
struct HeadTiles { static constexpr int DT = HDP == 96 ? 5 : HDP / 16; };

// ---- row-major image: 64 rows x HDP, 16-byte chunks XOR-swizzled --------------------------------------------
template <int HDP, int NT = 256>
struct RowTile {
  static constexpr int CHP = HDP / 8;
  static constexpr int NIT = (64 * CHP + NT - 1) / NT;
  static constexpr int BYTES = 64 * HDP * 2;
  static constexpr bool CAN_FULL = (64 * CHP) % NT == 0;
  // 64 complete rows with the full head dimension: no predicates at all
  static __device__ __forceinline__ void load_full(const bf16_t* __restrict__ base, int64_t rs, int r0, int tid,
                                                   u32x4_t* regs) {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int item = tid + it * NT;
      const int row = item / CHP, ch = item % CHP;
      regs[it] = *(const u32x4_t*)(base + (int64_t)(r0 + row) * rs + ch * 8);
    }
  }
  // base: pointer to element [row 0][col 0] of this (b,h) slice; rs: row stride in elements
  static __device__ __forceinline__ void load(const bf16_t* __restrict__ base, int64_t rs, int r0, int nrows, int hd,
                                              int tid, u32x4_t* regs) {
    if (r0 + 64 <= nrows && hd == HDP && CAN_FULL) {   // wave-uniform fast path: no per-item predicates
      load_full(base, rs, r0, tid, regs);
      return;
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int item = tid + it * NT;
      const int row = item / CHP, ch = item % CHP;
      u32x4_t v = {0, 0, 0, 0};
      if (item < 64 * CHP && r0 + row < nrows && ch * 8 < hd) v = *(const u32x4_t*)(base + (int64_t)(r0 + row) * rs + ch * 8);
      regs[it] = v;
    }
  }
  static __device__ __forceinline__ void store(char* lds, int tid, const u32x4_t* regs) {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int item = tid + it * NT;
      const int row = item / CHP, ch = item % CHP;
      if (item < 64 * CHP) *(u32x4_t*)(lds + row * (HDP * 2) + ((ch ^ rm_swz<HDP>(row)) * 16)) = regs[it];
    }
  }
  // MFMA fragment: 8 contiguous head-dim elements of row `row`, chunk index c
  static __device__ __forceinline__ bf16x8_t frag(const char* lds, int row, int c) {
    return *(const bf16x8_t*)(lds + row * (HDP * 2) + ((c ^ rm_swz<HDP>(row)) * 16));
  }
};

// ---- transposed fragments straight from the row-major image: ds_read_b64_tr_b16 ------------------------------
// MFMA operand whose contraction index is the TOKEN (V in P.V, Q/dO in dK/dV, K in dQ): lane (g = lane>>4,
// i = lane&15) needs column d0+i of the 8 tokens {t0+4g+0..3, t0+16+4g+0..3}.  The gfx950 transpose read delivers
// exactly that from row-major data: within a 16-lane group, lane 4j+q supplies the address of the 8-byte chunk
// (row j, columns 4q..4q+3) and lane i receives column i of the four rows (mapping verified on hardware by
// tests/test_rows_gpu.py::test_probe_tr16_dump).  Two reads (rows t0.. and t0+16..) fill the 8 k-slots.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
template <int HDP>
struct TrFrag {
  int row_off;   // byte offset of this lane's source row (t0 = 0)
  int kx;        // row & 7 (swizzle key of that row; identical for row+16)
  int qlo, qhi;  // (q & 1) * 8 and q >> 1 for this lane's 4-column chunk
  __device__ __forceinline__ TrFrag(int lane) {
    const int g = lane >> 4, j = (lane & 15) >> 2, q = lane & 3;
    const int row = 4 * g + j;
    row_off = row * (HDP * 2);
    kx = rm_swz<HDP>(row);
    qlo = (q & 1) * 8;
    qhi = q >> 1;
  }
  // tokens t0 + {4g..4g+3, 16+4g..16+4g+3} (t0 multiple of 32), columns d0 .. d0+15 (d0 multiple of 16)
  __device__ __forceinline__ bf16x8_t load(const char* lds, int t0, int d0) const {
    const char* p0 = lds + t0 * (HDP * 2) + row_off + ((((d0 >> 3) + qhi) ^ kx) * 16) + qlo;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 16 * HDP * 2));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, w);
  }
};

// ---- LDS-DMA staging of one 64-row tile of TWO row-major operands (X at +0, Y at +RowTile::BYTES of a ring buffer) ----
// Used by the backward kernels (the forward kernel carries its own copy of the same scheme).  Rows beyond `nrows`
// re-read row nrows-1 and chunks beyond the real head dimension re-read chunk 0: the callers mask such rows / never
// use such columns, so the DMA needs no zero fill.
template <int HDP, int NT>
struct TileDma {
  using RT = RowTile<HDP, NT>;
  static_assert(RT::CAN_FULL, "tile items must be a multiple of the workgroup size");
  static constexpr int NDMA = RT::NIT;
  int row[NDMA];
  unsigned col2[NDMA];   // byte offset of the (swizzled, clamped) source chunk inside a row
  int wu;
  __device__ __forceinline__ TileDma(int tid, int hd) {
    wu = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int it = 0; it < NDMA; it++) {
      const int item = tid + it * NT;
      const int r = item / RT::CHP, c = (item % RT::CHP) ^ rm_swz<HDP>(r);
      row[it] = r;
      col2[it] = (unsigned)(c * 8 < hd ? c * 8 : 0) * 2u;
    }
  }
  // wave-uniform tile bases in SGPRs + 32-bit per-lane byte offsets (row * row stride + chunk): no 64-bit vector address
  // arithmetic (these kernels are bound by VALU issue)
  template <bool FULL>
  __device__ __forceinline__ void issue(const bf16_t* xb, int64_t rsx, const bf16_t* yb, int64_t rsy, int r0, int nrows,
                                        char* buf) const {
    const bf16_t* xt = xb + (int64_t)r0 * rsx;   // uniform
    const bf16_t* yt = yb + (int64_t)r0 * rsy;
    const unsigned rsx2 = (unsigned)rsx * 2u, rsy2 = (unsigned)rsy * 2u;
    const int last = nrows - 1 - r0;              // >= 0: the tile exists
#pragma unroll
    for (int it = 0; it < NDMA; it++) {
      int r = row[it];
      if constexpr (!FULL) r = r < last ? r : last;
      char* dst = buf + (it * NT + wu * 64) * 16;   // wave-uniform; the hardware adds lane * 16
      dma16_sv(xt, (unsigned)r * rsx2 + col2[it], lds_addr(dst));
      dma16_sv(yt, (unsigned)r * rsy2 + col2[it], lds_addr(dst + RT::BYTES));
    }
  }
};

__device__ __forceinline__ bf16x8_t load_frag_global(const bf16_t* p, bool valid) {
  u32x4_t v = {0, 0, 0, 0};
  if (valid) v = *(const u32x4_t*)p;
  return __builtin_bit_cast(bf16x8_t, v);
}

// Several [B_i, S_i] segments of one token-major activation (the two masks of a V-JEPA batch) in ONE launch: the workgroups of
// the short segment fill the tail of the long one instead of paying a launch of their own (ViT-L context encoder: 376- and
// 112-token segments, the second alone runs at 90-280 TF/s).  Workgroup -> (segment, local index) by the cumulative counts.
#define VJ_ATTN_MAX_SEGS 4
struct AttnSegs {
  int n;
  int blk_end[VJ_ATTN_MAX_SEGS];    // cumulative workgroup counts
  int S[VJ_ATTN_MAX_SEGS];          // sequence length
  int nb[VJ_ATTN_MAX_SEGS];         // query blocks (forward, dQ) or key blocks (dK/dV) per (sample, head)
  int64_t row0[VJ_ATTN_MAX_SEGS];   // first token row of the segment
  int64_t col0[VJ_ATTN_MAX_SEGS];   // first column-partial row of the segment (backward with column sums)
};
__device__ __forceinline__ int attn_seg_of(const AttnSegs& sg, int& logical) {
  int si = 0;
#pragma unroll
  for (int i = 0; i + 1 < VJ_ATTN_MAX_SEGS; i++)
    if (i + 1 < sg.n && logical >= sg.blk_end[i]) si = i + 1;
  logical -= si > 0 ? sg.blk_end[si - 1] : 0;
  return si;
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int qx = nblk >> 3, rx = nblk & 7, xcd = bid & 7, pos = bid >> 3;
  return (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + pos;
}

