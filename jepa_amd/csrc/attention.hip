// Dense (non-causal, unmasked) multi-head attention over mask-gathered token sequences, forward and backward,
// for gfx950.  Reference call site: F.scaled_dot_product_attention(q, k, v) in Attention.forward
// (src/models/utils/modules.py:61-78); the `mask=` argument there is accepted and ignored, so "masked attention"
// is dense attention over the shortened (gathered) sequence, with default scale head_dim**-0.5.
//
// q/k/v are read strided straight out of the packed qkv GEMM output [B, S, 3, H, hd] (no permute copies), the
// output is written token-major [B, S, H*hd] ready for the proj GEMM; the backward writes dq/dk/dv packed as
// dqkv [B, S, 3, H, hd] ready for the qkv dgrad/wgrad GEMMs.
//
// All three kernels compute TRANSPOSED products with MFMA 16x16x32 so that (i) softmax rows / per-query scalars
// are lane-local (query = lane&15), (ii) the exponentiated scores feed the next MFMA straight from the
// accumulator registers (the contraction index is permuted identically on both operands, so no cross-lane
// movement), and (iii) every lane owns 4 consecutive head-dim columns of the result (8-byte stores).
// K/V (or Q/dO) tiles of 64 rows are staged through LDS ONCE, row-major with a 16-byte XOR swizzle: operands that
// contract over head-dim are read as ds_read_b128 fragments, operands that contract over the token index are read
// with the gfx950 transpose read ds_read_b64_tr_b16 from the same image (no transposed copy, no scatter stores).
// The next tile's global loads are issued before the current tile's MFMAs.
//
// Softmax is computed in base 2: s2 = (q.k) * scale * log2(e); lse2 = max2 + log2(sum) is what the forward
// saves for the backward (an internal format, produced and consumed only here).
#include "common.hpp"


__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  bf2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

#define DEFER_LOG2 5.0f  // forward softmax: rescale O only when a row max grows by more than 2^5

template <int HDP>
__device__ __forceinline__ int rm_swz(int row) {
  return HDP >= 64 ? (row & 7) : 0;
}

// ---- row-major image: 64 rows x HDP, 16-byte chunks XOR-swizzled --------------------------------------------
template <int HDP, int NT = 256>
struct RowTile {
  static constexpr int CHP = HDP / 8;
  static constexpr int NIT = (64 * CHP + NT - 1) / NT;
  static constexpr int BYTES = 64 * HDP * 2;
  // base: pointer to element [row 0][col 0] of this (b,h) slice; rs: row stride in elements
  static __device__ __forceinline__ void load(const bf16_t* __restrict__ base, int64_t rs, int r0, int nrows, int hd,
                                              int tid, u32x4_t* regs) {
    if (r0 + 64 <= nrows && hd == HDP && (64 * CHP) % NT == 0) {   // wave-uniform fast path: no per-item predicates
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int item = tid + it * NT;
        const int row = item / CHP, ch = item % CHP;
        regs[it] = *(const u32x4_t*)(base + (int64_t)(r0 + row) * rs + ch * 8);
      }
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
// tests/test_kernels_gpu.py::test_probe_tr16_dump).  Two reads (rows t0.. and t0+16..) fill the 8 k-slots.
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

__device__ __forceinline__ bf16x8_t load_frag_global(const bf16_t* p, bool valid) {
  u32x4_t v = {0, 0, 0, 0};
  if (valid) v = *(const u32x4_t*)p;
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int qx = nblk >> 3, rx = nblk & 7, xcd = bid & 7, pos = bid >> 3;
  return (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + pos;
}

// =============================================================================================================
// forward:  O = softmax(Q K^T * scale) V,  128 queries per workgroup (32 per wave), 64-key tiles
// =============================================================================================================
template <int HDP, int QT>
__global__ __launch_bounds__(8 * 64 / QT) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                       float* __restrict__ lse2, int B, int S, int H, int hd,
                                                       float sc, int nqb) {
  constexpr int NT = 8 * 64 / QT;   // 128 queries per workgroup, 16*QT per wave
  using RT = RowTile<HDP, NT>;
  __shared__ __attribute__((aligned(16))) char smem[4 * RT::BYTES];   // {K,V} x 2 buffers
  const TrFrag<HDP> trf(threadIdx.x & 63);
  constexpr int KS = HDP / 32, DT = HDP / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = logical % nqb, bh = logical / nqb;
  const int h = bh % H, b = bh / H;
  const int64_t rs = (int64_t)3 * H * hd;
  const bf16_t* qbase = qkv + (int64_t)b * S * rs + (int64_t)h * hd;
  const bf16_t* kbase = qbase + (int64_t)H * hd;
  const bf16_t* vbase = qbase + (int64_t)2 * H * hd;
  const int q0 = qb * 128 + w * (16 * QT);

  bf16x8_t qf[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; qt++)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int q = q0 + qt * 16 + li, d0 = ks * 32 + 8 * g;
      qf[qt][ks] = load_frag_global(qbase + (int64_t)q * rs + d0, q < S && d0 < hd);
    }

  f32x4_t oacc[QT][DT];
#pragma unroll
  for (int qt = 0; qt < QT; qt++)
#pragma unroll
    for (int dt = 0; dt < DT; dt++) oacc[qt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float mrun[QT], lrun[QT];
#pragma unroll
  for (int qt = 0; qt < QT; qt++) {
    mrun[qt] = -INFINITY;
    lrun[qt] = 0.f;
  }

  u32x4_t kreg[RT::NIT], vreg[RT::NIT];
  const int nt = (S + 63) / 64;
  // double-buffered LDS image, ONE barrier per tile: tile t+1 is written into the other buffer right after the
  // barrier of tile t (its global loads were issued a whole tile earlier), tile t+2's loads are issued next.
  RT::load(kbase, rs, 0, S, hd, tid, kreg);
  RT::load(vbase, rs, 0, S, hd, tid, vreg);
  RT::store(smem, tid, kreg);
  RT::store(smem + RT::BYTES, tid, vreg);
  if (nt > 1) {
    RT::load(kbase, rs, 64, S, hd, tid, kreg);
    RT::load(vbase, rs, 64, S, hd, tid, vreg);
  }

  for (int t = 0; t < nt; t++) {
    const int k0 = t * 64;
    char* k_lds = smem + (t & 1) * (2 * RT::BYTES);
    char* v_lds = k_lds + RT::BYTES;
    __syncthreads();  // tile t is in LDS for every wave; every wave is done reading tile t-1 (the other buffer)
    if (t + 1 < nt) {
      char* kn = smem + ((t + 1) & 1) * (2 * RT::BYTES);
      RT::store(kn, tid, kreg);
      RT::store(kn + RT::BYTES, tid, vreg);
      if (t + 2 < nt) {
        RT::load(kbase, rs, k0 + 128, S, hd, tid, kreg);
        RT::load(vbase, rs, k0 + 128, S, hd, tid, vreg);
      }
    }
    // ---- S^T = K Q^T : sacc[qt][kt] holds S^T[key = kt*16 + 4g + r][q = li] ----
    f32x4_t sacc[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; qt++)
#pragma unroll
      for (int kt = 0; kt < 4; kt++) sacc[qt][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; kt++)
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const bf16x8_t kf = RT::frag(k_lds, kt * 16 + li, ks * 4 + g);
#pragma unroll
        for (int qt = 0; qt < QT; qt++)
          sacc[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], sacc[qt][kt], 0, 0, 0);
      }
    // ---- online softmax (query = li, reduced over r, kt in-lane and over g across lanes 16/32 apart) ----
    bf16x8_t pf[QT][2];
    const bool tail = (k0 + 64 > S);  // wave-uniform: only the last tile can hold padded keys
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
      if (tail) {
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (k0 + kt * 16 + 4 * g + r >= S) sacc[qt][kt][r] = -INFINITY;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) mx = fmaxf(mx, sacc[qt][kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // deferred rescale: keep the old running max unless some row of this wave grew by more than 2^DEFER_LOG2;
      // P is then bounded by 2^DEFER_LOG2 instead of 1 (bf16 rounding is scale-free, fp32 accumulators have the
      // headroom), and the O / l rescale -- a full pass over the accumulators -- is skipped on most tiles.
      const float mcand = mx * sc;   // sc > 0: max commutes with the scaling
      float mnew = mrun[qt];
      if (__any(mcand > mrun[qt] + DEFER_LOG2)) {   // wave-uniform; first tile always (mrun = -inf)
        mnew = fmaxf(mrun[qt], mcand);
        const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
        lrun[qt] *= alpha;
        mrun[qt] = mnew;
#pragma unroll
        for (int dt = 0; dt < DT; dt++) oacc[qt][dt] *= alpha;
      }
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float p = __builtin_amdgcn_exp2f(fmaf(sacc[qt][kt][r], sc, -mnew));
          sacc[qt][kt][r] = p;
          ls += p;
        }
      lrun[qt] += ls;
#pragma unroll
      for (int c = 0; c < 2; c++) {
        u32x4_t pw;
        pw[0] = cvt_pk_bf16(sacc[qt][2 * c][0], sacc[qt][2 * c][1]);
        pw[1] = cvt_pk_bf16(sacc[qt][2 * c][2], sacc[qt][2 * c][3]);
        pw[2] = cvt_pk_bf16(sacc[qt][2 * c + 1][0], sacc[qt][2 * c + 1][1]);
        pw[3] = cvt_pk_bf16(sacc[qt][2 * c + 1][2], sacc[qt][2 * c + 1][3]);
        pf[qt][c] = __builtin_bit_cast(bf16x8_t, pw);
      }
    }
    // ---- O^T += V^T P^T : oacc[qt][dt] holds O^T[d = dt*16 + 4g + r][q = li] ----
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const bf16x8_t vf = trf.load(v_lds, c * 32, dt * 16);
#pragma unroll
        for (int qt = 0; qt < QT; qt++)
          oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][c], oacc[qt][dt], 0, 0, 0);
      }
  }

#pragma unroll
  for (int qt = 0; qt < QT; qt++) {
    float l = lrun[qt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const int q = q0 + qt * 16 + li;
    if (q < S) {
      const float inv = 1.0f / l;
      bf16_t* op = o + ((int64_t)b * S + q) * ((int64_t)H * hd) + (int64_t)h * hd;
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const int d = dt * 16 + 4 * g;
        if (d < hd) {
          u32x2_t ov;
          ov[0] = cvt_pk_bf16(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv);
          ov[1] = cvt_pk_bf16(oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv);
          *(u32x2_t*)(op + d) = ov;
        }
      }
      if (g == 0 && lse2) lse2[((int64_t)b * H + h) * S + q] = mrun[qt] + log2f(l);
    }
  }
}

// =============================================================================================================
// delta[b,h,s] = sum_d dO[b,s,h,d] * O[b,s,h,d]     (softmax-backward row term)
// =============================================================================================================
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                         float* __restrict__ delta, int64_t B, int S, int H, int hd) {
  const int64_t total = B * S * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int h = (int)(i % H);
    const int64_t bs = i / H;
    const int s = (int)(bs % S);
    const int64_t b = bs / S;
    const bf16_t* op = o + i * hd;
    const bf16_t* dp = dout + i * hd;
    float acc = 0.f;
    for (int d = 0; d < hd; d += 8) {
      const u32x4_t a = *(const u32x4_t*)(op + d);
      const u32x4_t c = *(const u32x4_t*)(dp + d);
#pragma unroll
      for (int j = 0; j < 4; j++) acc += bf_lo(a[j]) * bf_lo(c[j]) + bf_hi(a[j]) * bf_hi(c[j]);
    }
    delta[(b * H + h) * S + s] = acc;
  }
}

// =============================================================================================================
// backward, part 1: dK, dV.  One workgroup per 64-key block (16 keys per wave), loop over 64-query tiles.
//   S = Q K^T (lane: S[q = qt*16+4g+r][key = li]),  P = exp2(S*sc - lse2[q]),  dP = dO V^T,
//   dS = P (dP - delta[q]),  dV^T += dO^T P,  dK^T += Q^T dS  (then * scale)
// =============================================================================================================
template <int HDP>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(const bf16_t* __restrict__ qkv,
                                                            const bf16_t* __restrict__ dout,
                                                            const float* __restrict__ lse2,
                                                            const float* __restrict__ delta,
                                                            bf16_t* __restrict__ dqkv, int B, int S, int H, int hd,
                                                            float sc, float scale, int nkb) {
  __shared__ __attribute__((aligned(16))) char smem[2 * RowTile<HDP>::BYTES + 2 * 64 * 4];
  char* q_lds = smem;
  char* do_lds = smem + RowTile<HDP>::BYTES;
  float* lse_s = (float*)(smem + 2 * RowTile<HDP>::BYTES);
  float* dl_s = lse_s + 64;
  const TrFrag<HDP> trf(threadIdx.x & 63);
  constexpr int KS = HDP / 32, DT = HDP / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int kb = logical % nkb, bh = logical / nkb;
  const int h = bh % H, b = bh / H;
  const int64_t rs = (int64_t)3 * H * hd;
  const int64_t os = (int64_t)H * hd;
  const bf16_t* qbase = qkv + (int64_t)b * S * rs + (int64_t)h * hd;
  const bf16_t* kbase = qbase + (int64_t)H * hd;
  const bf16_t* vbase = qbase + (int64_t)2 * H * hd;
  const bf16_t* dobase = dout + (int64_t)b * S * os + (int64_t)h * hd;
  const float* lse_b = lse2 + ((int64_t)b * H + h) * S;
  const float* dl_b = delta + ((int64_t)b * H + h) * S;
  const int key = kb * 64 + w * 16 + li;
  const bool key_ok = key < S;

  bf16x8_t kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    const int d0 = ks * 32 + 8 * g;
    kf[ks] = load_frag_global(kbase + (int64_t)key * rs + d0, key_ok && d0 < hd);
    vf[ks] = load_frag_global(vbase + (int64_t)key * rs + d0, key_ok && d0 < hd);
  }
  f32x4_t dvacc[DT], dkacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; dt++) {
    dvacc[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    dkacc[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  u32x4_t qreg[RowTile<HDP>::NIT], doreg[RowTile<HDP>::NIT];
  float lse_r = 0.f, dl_r = 0.f;
  const int nt = (S + 63) / 64;
  auto load_tile = [&](int r0) {
    RowTile<HDP>::load(qbase, rs, r0, S, hd, tid, qreg);
    RowTile<HDP>::load(dobase, os, r0, S, hd, tid, doreg);
    if (tid < 64) {
      const int q = r0 + tid;
      lse_r = q < S ? lse_b[q] : INFINITY;  // +inf -> P = 0 for padded query rows
      dl_r = q < S ? dl_b[q] : 0.f;
    }
  };
  load_tile(0);

  for (int t = 0; t < nt; t++) {
    __syncthreads();
    RowTile<HDP>::store(q_lds, tid, qreg);
    RowTile<HDP>::store(do_lds, tid, doreg);
    if (tid < 64) {
      lse_s[tid] = lse_r;
      dl_s[tid] = dl_r;
    }
    __syncthreads();
    if (t + 1 < nt) load_tile((t + 1) * 64);

    float pv[4][4], dsv[4][4];
#pragma unroll
    for (int qt = 0; qt < 4; qt++) {
      f32x4_t sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const bf16x8_t qa = RowTile<HDP>::frag(q_lds, qt * 16 + li, ks * 4 + g);
        const bf16x8_t da = RowTile<HDP>::frag(do_lds, qt * 16 + li, ks * 4 + g);
        sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ks], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[ks], dpacc, 0, 0, 0);
      }
      const float4 lse4 = *(const float4*)(lse_s + qt * 16 + 4 * g);   // queries qt*16 + 4g + {0..3}
      const float4 dl4 = *(const float4*)(dl_s + qt * 16 + 4 * g);
      const float lse_a[4] = {lse4.x, lse4.y, lse4.z, lse4.w}, dl_a[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[r], sc, -lse_a[r]));  // padded queries: lse = +inf -> 0
        pv[qt][r] = p;
        dsv[qt][r] = p * (dpacc[r] - dl_a[r]);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      u32x4_t pw, dw;
      pw[0] = cvt_pk_bf16(pv[2 * c][0], pv[2 * c][1]);
      pw[1] = cvt_pk_bf16(pv[2 * c][2], pv[2 * c][3]);
      pw[2] = cvt_pk_bf16(pv[2 * c + 1][0], pv[2 * c + 1][1]);
      pw[3] = cvt_pk_bf16(pv[2 * c + 1][2], pv[2 * c + 1][3]);
      dw[0] = cvt_pk_bf16(dsv[2 * c][0], dsv[2 * c][1]);
      dw[1] = cvt_pk_bf16(dsv[2 * c][2], dsv[2 * c][3]);
      dw[2] = cvt_pk_bf16(dsv[2 * c + 1][0], dsv[2 * c + 1][1]);
      dw[3] = cvt_pk_bf16(dsv[2 * c + 1][2], dsv[2 * c + 1][3]);
      const bf16x8_t pfr = __builtin_bit_cast(bf16x8_t, pw);
      const bf16x8_t dfr = __builtin_bit_cast(bf16x8_t, dw);
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const bf16x8_t dot_f = trf.load(do_lds, c * 32, dt * 16);
        const bf16x8_t qt_f = trf.load(q_lds, c * 32, dt * 16);
        dvacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_f, pfr, dvacc[dt], 0, 0, 0);
        dkacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f, dfr, dkacc[dt], 0, 0, 0);
      }
    }
  }

  if (key_ok) {
    bf16_t* dkp = dqkv + ((int64_t)b * S + key) * rs + (int64_t)H * hd + (int64_t)h * hd;
    bf16_t* dvp = dkp + (int64_t)H * hd;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
      const int d = dt * 16 + 4 * g;
      if (d < hd) {
        u32x2_t a, c;
        a[0] = cvt_pk_bf16(dkacc[dt][0] * scale, dkacc[dt][1] * scale);
        a[1] = cvt_pk_bf16(dkacc[dt][2] * scale, dkacc[dt][3] * scale);
        c[0] = cvt_pk_bf16(dvacc[dt][0], dvacc[dt][1]);
        c[1] = cvt_pk_bf16(dvacc[dt][2], dvacc[dt][3]);
        *(u32x2_t*)(dkp + d) = a;
        *(u32x2_t*)(dvp + d) = c;
      }
    }
  }
}

// =============================================================================================================
// backward, part 2: dQ.  One workgroup per 128 queries (32 per wave), loop over 64-key tiles.
//   S^T = K Q^T, dP^T = V dO^T (lane: [key = kt*16+4g+r][q = li]),  dS^T = P^T (dP^T - delta[q]),
//   dQ^T += K^T dS^T  (then * scale)
// =============================================================================================================
template <int HDP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv,
                                                          const bf16_t* __restrict__ dout,
                                                          const float* __restrict__ lse2,
                                                          const float* __restrict__ delta,
                                                          bf16_t* __restrict__ dqkv, int B, int S, int H, int hd,
                                                          float sc, float scale, int nqb) {
  __shared__ __attribute__((aligned(16))) char smem[2 * RowTile<HDP>::BYTES];
  char* k_lds = smem;
  char* v_lds = smem + RowTile<HDP>::BYTES;
  const TrFrag<HDP> trf(threadIdx.x & 63);
  constexpr int KS = HDP / 32, DT = HDP / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = logical % nqb, bh = logical / nqb;
  const int h = bh % H, b = bh / H;
  const int64_t rs = (int64_t)3 * H * hd;
  const int64_t os = (int64_t)H * hd;
  const bf16_t* qbase = qkv + (int64_t)b * S * rs + (int64_t)h * hd;
  const bf16_t* kbase = qbase + (int64_t)H * hd;
  const bf16_t* vbase = qbase + (int64_t)2 * H * hd;
  const bf16_t* dobase = dout + (int64_t)b * S * os + (int64_t)h * hd;
  const int q0 = qb * 128 + w * 32;

  bf16x8_t qf[2][KS], dof[2][KS];
  float lse_q[2], dl_q[2];
#pragma unroll
  for (int qt = 0; qt < 2; qt++) {
    const int q = q0 + qt * 16 + li;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int d0 = ks * 32 + 8 * g;
      qf[qt][ks] = load_frag_global(qbase + (int64_t)q * rs + d0, q < S && d0 < hd);
      dof[qt][ks] = load_frag_global(dobase + (int64_t)q * os + d0, q < S && d0 < hd);
    }
    lse_q[qt] = q < S ? lse2[((int64_t)b * H + h) * S + q] : INFINITY;
    dl_q[qt] = q < S ? delta[((int64_t)b * H + h) * S + q] : 0.f;
  }
  f32x4_t dqacc[2][DT];
#pragma unroll
  for (int qt = 0; qt < 2; qt++)
#pragma unroll
    for (int dt = 0; dt < DT; dt++) dqacc[qt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  u32x4_t kreg[RowTile<HDP>::NIT], vreg[RowTile<HDP>::NIT];
  const int nt = (S + 63) / 64;
  RowTile<HDP>::load(kbase, rs, 0, S, hd, tid, kreg);
  RowTile<HDP>::load(vbase, rs, 0, S, hd, tid, vreg);

  for (int t = 0; t < nt; t++) {
    const int k0 = t * 64;
    __syncthreads();
    RowTile<HDP>::store(k_lds, tid, kreg);
    RowTile<HDP>::store(v_lds, tid, vreg);
    __syncthreads();
    if (t + 1 < nt) {
      RowTile<HDP>::load(kbase, rs, k0 + 64, S, hd, tid, kreg);
      RowTile<HDP>::load(vbase, rs, k0 + 64, S, hd, tid, vreg);
    }
    f32x4_t sacc[2][4], dpacc[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; qt++)
#pragma unroll
      for (int kt = 0; kt < 4; kt++) {
        sacc[qt][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        dpacc[qt][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int kt = 0; kt < 4; kt++)
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const bf16x8_t ka = RowTile<HDP>::frag(k_lds, kt * 16 + li, ks * 4 + g);
        const bf16x8_t va = RowTile<HDP>::frag(v_lds, kt * 16 + li, ks * 4 + g);
#pragma unroll
        for (int qt = 0; qt < 2; qt++) {
          sacc[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[qt][ks], sacc[qt][kt], 0, 0, 0);
          dpacc[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, dof[qt][ks], dpacc[qt][kt], 0, 0, 0);
        }
      }
    bf16x8_t dsf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
#pragma unroll
      for (int kt = 0; kt < 4; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float p = __builtin_amdgcn_exp2f(fmaf(sacc[qt][kt][r], sc, -lse_q[qt]));
          sacc[qt][kt][r] = p * (dpacc[qt][kt][r] - dl_q[qt]);
        }
#pragma unroll
      for (int c = 0; c < 2; c++) {
        u32x4_t dw;
        dw[0] = cvt_pk_bf16(sacc[qt][2 * c][0], sacc[qt][2 * c][1]);
        dw[1] = cvt_pk_bf16(sacc[qt][2 * c][2], sacc[qt][2 * c][3]);
        dw[2] = cvt_pk_bf16(sacc[qt][2 * c + 1][0], sacc[qt][2 * c + 1][1]);
        dw[3] = cvt_pk_bf16(sacc[qt][2 * c + 1][2], sacc[qt][2 * c + 1][3]);
        dsf[qt][c] = __builtin_bit_cast(bf16x8_t, dw);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const bf16x8_t ktf = trf.load(k_lds, c * 32, dt * 16);
#pragma unroll
        for (int qt = 0; qt < 2; qt++)
          dqacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qt][c], dqacc[qt][dt], 0, 0, 0);
      }
  }

#pragma unroll
  for (int qt = 0; qt < 2; qt++) {
    const int q = q0 + qt * 16 + li;
    if (q < S) {
      bf16_t* dqp = dqkv + ((int64_t)b * S + q) * rs + (int64_t)h * hd;
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const int d = dt * 16 + 4 * g;
        if (d < hd) {
          u32x2_t a;
          a[0] = cvt_pk_bf16(dqacc[qt][dt][0] * scale, dqacc[qt][dt][1] * scale);
          a[1] = cvt_pk_bf16(dqacc[qt][dt][2] * scale, dqacc[qt][dt][3] * scale);
          *(u32x2_t*)(dqp + d) = a;
        }
      }
    }
  }
}

// =============================================================================================================
// host entry points
// =============================================================================================================
static int g_attn_fwd_qt = 2;   // A/B switch for tools/attn_bench.py (vj_attn_set_variant)
extern "C" int vj_attn_set_variant(int fwd_qt) {
  g_attn_fwd_qt = (fwd_qt == 1) ? 1 : 2;
  return 0;
}

static int pick_hdp(int64_t hd) { return hd <= 32 ? 32 : (hd <= 64 ? 64 : (hd <= 128 ? 128 : 0)); }
#define LOG2E 1.4426950408889634f

extern "C" int vj_attn_fwd(const void* qkv, void* o, float* lse2, int64_t B, int64_t S, int64_t H, int64_t hd,
                           float scale, hipStream_t stream) {
  VJ_CHECK_ARG(hd % 8 == 0 && pick_hdp(hd) != 0, "vj_attn_fwd: head_dim=%ld unsupported (need %%8==0, <=128)", (long)hd);
  VJ_CHECK_ARG(B >= 0 && S >= 0 && H > 0, "vj_attn_fwd: bad dims");
  if (B * S == 0) return 0;
  const int nqb = (int)cdiv64(S, 128);
  const int64_t nblk = B * H * nqb;
  VJ_CHECK_ARG(nblk < (1ll << 31), "vj_attn_fwd: grid too large");
  const float sc = scale * LOG2E;
  // QT = 16-row query tiles per wave (QT = 1: 8 waves / workgroup, half the registers per wave).  Measured on
  // MI355X: QT = 2 is faster or equal for every head size (the kernel is bound by VALU issue, not by occupancy).
#define VJ_FWD(HDPV, QTV)                                                                                            \
  hipLaunchKernelGGL((attn_fwd_kernel<HDPV, QTV>), dim3((unsigned)nblk), dim3(8 * 64 / QTV), 0, stream,                \
                     (const bf16_t*)qkv, (bf16_t*)o, lse2, (int)B, (int)S, (int)H, (int)hd, sc, nqb)
  const int qt_sel = g_attn_fwd_qt;
  switch (pick_hdp(hd)) {
    case 32:
      if (qt_sel == 2) VJ_FWD(32, 2); else VJ_FWD(32, 1);
      break;
    case 64:
      if (qt_sel == 2) VJ_FWD(64, 2); else VJ_FWD(64, 1);
      break;
    default:
      if (qt_sel == 2) VJ_FWD(128, 2); else VJ_FWD(128, 1);
  }
#undef VJ_FWD
  VJ_LAUNCH_CHECK("vj_attn_fwd");
  return 0;
}

extern "C" int64_t vj_attn_bwd_ws_bytes(int64_t B, int64_t S, int64_t H) { return B * S * H * 4; }

extern "C" int vj_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse2, void* dqkv,
                           int64_t B, int64_t S, int64_t H, int64_t hd, float scale, void* ws, int64_t ws_bytes,
                           hipStream_t stream) {
  VJ_CHECK_ARG(hd % 8 == 0 && pick_hdp(hd) != 0, "vj_attn_bwd: head_dim=%ld unsupported", (long)hd);
  VJ_CHECK_ARG(ws_bytes >= vj_attn_bwd_ws_bytes(B, S, H), "vj_attn_bwd: workspace too small");
  if (B * S == 0) return 0;
  float* delta = (float*)ws;
  {
    int64_t gsz = cdiv64(B * S * H, 256);
    if (gsz > 256 * 16) gsz = 256 * 16;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)gsz), dim3(256), 0, stream, (const bf16_t*)o,
                       (const bf16_t*)dout, delta, B, (int)S, (int)H, (int)hd);
    VJ_LAUNCH_CHECK("vj_attn_bwd(delta)");
  }
  const int nkb = (int)cdiv64(S, 64), nqb = (int)cdiv64(S, 128);
  const int64_t g1 = B * H * nkb, g2 = B * H * nqb;
  VJ_CHECK_ARG(g1 < (1ll << 31), "vj_attn_bwd: grid too large");
  const float sc = scale * LOG2E;
#define VJ_BWD_LAUNCH(HDPV)                                                                                        \
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel<HDPV>, dim3((unsigned)g1), dim3(256), 0, stream, (const bf16_t*)qkv,     \
                     (const bf16_t*)dout, lse2, delta, (bf16_t*)dqkv, (int)B, (int)S, (int)H, (int)hd, sc, scale,   \
                     nkb);                                                                                         \
  hipLaunchKernelGGL(attn_bwd_dq_kernel<HDPV>, dim3((unsigned)g2), dim3(256), 0, stream, (const bf16_t*)qkv,       \
                     (const bf16_t*)dout, lse2, delta, (bf16_t*)dqkv, (int)B, (int)S, (int)H, (int)hd, sc, scale,   \
                     nqb);
  switch (pick_hdp(hd)) {
    case 32: VJ_BWD_LAUNCH(32); break;
    case 64: VJ_BWD_LAUNCH(64); break;
    default: VJ_BWD_LAUNCH(128);
  }
#undef VJ_BWD_LAUNCH
  VJ_LAUNCH_CHECK("vj_attn_bwd");
  return 0;
}
