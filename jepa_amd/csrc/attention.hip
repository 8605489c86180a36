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
#include "attn_common.hpp"

// =============================================================================================================
// forward:  O = softmax(Q K^T * scale) V,  128 queries per workgroup (32 per wave), 64-key tiles; "seeded" soft-max (SM_HEADROOM in
// attn_common.hpp).  (The round-3 kernel with a per-tile row maximum and a per-score FMA is gone: profiles/r04_abab_attention_biasfuse.md)
// =============================================================================================================
// RS = where the soft-max row sums come from: 0 vector adds (v_pk_add_f32 on the fp32 probabilities), 1 the V pad column (head_dim 24),
// 2 an all-ones A operand: two extra P MFMAs per key tile and 16-row block put sum_k bf16(P) in every row of a 16 x 16 accumulator --
// packed f32 adds next to MFMAs cost twice their stand-alone time (profiles/r03_valu_mfma_probe.md), the two MFMAs 32 cycles
template <int HDP, int QT, int NBUF, int RS>
__global__ __launch_bounds__(8 * 64 / QT) void attn_fwd_sm_kernel(const bf16_t* __restrict__ qkv_all,
                                                          bf16_t* __restrict__ o_all, float* __restrict__ lse2_all,
                                                          AttnSegs sg, int H, int hd, float sc) {
  constexpr int NT = 8 * 64 / QT;
  using RT = RowTile<HDP, NT>;
  constexpr int DIST = NBUF - 1;
  static_assert(RT::CAN_FULL || NT == 512, "tile items must be a multiple of the workgroup size");
  constexpr bool PSUM = RS == 1;
  static_assert(!PSUM || (HDP == 32 && RT::CAN_FULL && RT::NIT == 1), "row sums on the pad column: 32-wide class, one DMA item per thread");
  constexpr int NDMA = RT::CAN_FULL ? RT::NIT : 1;
  __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * RT::BYTES];
  const TrFrag<HDP> trf(threadIdx.x & 63);
  constexpr int KS = HDP / 32, DT = HeadTiles<HDP>::DT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int si = attn_seg_of(sg, logical);
  const int S = sg.S[si], nqb = sg.nb[si];
  const int64_t rs = (int64_t)3 * H * hd;
  const bf16_t* qkv = qkv_all + sg.row0[si] * rs;
  bf16_t* o = o_all + sg.row0[si] * ((int64_t)H * hd);
  float* lse2 = lse2_all ? lse2_all + (int64_t)H * sg.row0[si] : nullptr;
  const int qb = logical % nqb, bh = logical / nqb;
  const int h = bh % H, b = bh / H;
  const bf16_t* qbase = qkv + (int64_t)b * S * rs + (int64_t)h * hd;
  const bf16_t* kbase = qbase + (int64_t)H * hd;
  const int q0 = qb * 128 + w * (16 * QT);

  // Q fragments carry the soft-max scale: bf16(q * scale * log2 e), once per workgroup
  bf16x8_t qf[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; qt++)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int q = q0 + qt * 16 + li, d0 = ks * 32 + 8 * g;
      qf[qt][ks] = scale_frag(load_frag_global(qbase + (int64_t)q * rs + d0, q < S && d0 < hd), sc);
    }

  f32x4_t oacc[QT][DT];
#pragma unroll
  for (int qt = 0; qt < QT; qt++)
#pragma unroll
    for (int dt = 0; dt < DT; dt++) oacc[qt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // mrun: the base the scores are measured against (s - mrun comes out of the matrix pipe); 0 until the first tile set it
  float mrun[QT], lrun[QT];
  f32x4_t lacc[QT];            // RS = 2: row sums out of the matrix pipe (every row of the block holds sum_k P[q = li][k])
  f32x4_t seed[QT];            // {-mrun} x 4: srcC of the first score MFMA of every key tile
#pragma unroll
  for (int qt = 0; qt < QT; qt++) {
    mrun[qt] = 0.f;
    lrun[qt] = 0.f;
    lacc[qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    seed[qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  const int nt = (S + 63) / 64;
  int dma_row[NDMA];
  unsigned dma_voff[NDMA], dma_col2[NDMA];
  const bool dma_on = RT::CAN_FULL || tid < 64 * RT::CHP;
  bool vpad = false;           // PSUM: this thread's DMA item is the pad chunk (columns >= hd) of its V row
#pragma unroll
  for (int it = 0; it < NDMA; it++) {
    const int item = RT::CAN_FULL ? tid + it * NT : (tid < 64 * RT::CHP ? tid : 0);
    const int row = item / RT::CHP, c = (item % RT::CHP) ^ rm_swz<HDP>(row);
    const int col = c * 8 < hd ? c * 8 : 0;
    if constexpr (PSUM) vpad = c * 8 >= hd;
    dma_row[it] = row;
    dma_col2[it] = (unsigned)col * 2u;
    dma_voff[it] = ((unsigned)row * (unsigned)rs + (unsigned)col) * 2u;
  }
  if constexpr (PSUM) {
    // the pad chunk of every V row: column hd = 1.0, the rest 0, written ONCE into every ring buffer; the V DMA skips these
    // lanes (exec-masked), so the chunk survives every tile.  P.V then accumulates sum_k P[q][k] in output column hd.
    if (vpad) {
      const u32x4_t one = {0x00003F80u, 0u, 0u, 0u};
#pragma unroll
      for (int d = 0; d < NBUF; d++) *(u32x4_t*)(smem + d * 2 * RT::BYTES + RT::BYTES + tid * 16) = one;
    }
  }
  const int64_t v_off = (int64_t)H * hd;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const unsigned rs2 = (unsigned)rs * 2u;
  auto issue = [&](const int tile, const int buf_off, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    char* kb = smem + buf_off;
    const bf16_t* kt = kbase + (int64_t)tile * 64 * rs;   // uniform
    const bf16_t* vt = kt + v_off;
#pragma unroll
    for (int it = 0; it < NDMA; it++) {
      unsigned vo;
      if constexpr (FULL) {
        vo = dma_voff[it];
      } else {
        const int last = S - 1 - tile * 64;
        const int r = dma_row[it] < last ? dma_row[it] : last;
        vo = (unsigned)r * rs2 + dma_col2[it];
      }
      char* dst = kb + (it * NT + wu * 64) * 16;
      if (dma_on) {
        dma16_sv(kt, vo, lds_addr(dst));
        if constexpr (PSUM) {
          if (!vpad) dma16_sv(vt, vo, lds_addr(dst + RT::BYTES));   // never all lanes of a wave: one pad chunk per 4 lanes
        } else {
          dma16_sv(vt, vo, lds_addr(dst + RT::BYTES));
        }
      }
    }
  };
  constexpr int BUFB = 2 * RT::BYTES, RINGB = NBUF * BUFB;
  __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
  for (int d = 0; d < DIST; d++)
    if (d < nt) issue(d, d * BUFB, std::false_type{});
  int cur_off = 0, nxt_off = DIST * BUFB;
  const bool wave_live = qb * 128 + wu * (16 * QT) < S;   // wave-uniform

  // FIRST: tile 0 -- the base is unknown (seed 0), the exact-maximum path runs unconditionally and nothing is rescaled.
  auto iter = [&](const int t, auto fast_tag, auto first_tag) {
    constexpr bool FAST = decltype(fast_tag)::value, FIRST = decltype(first_tag)::value;
    const int k0 = t * 64;
    char* k_lds = smem + cur_off;
    char* v_lds = k_lds + RT::BYTES;
    if (DIST >= 2 && (FAST || t + 1 < nt)) wait_vmcnt<2 * NDMA>();
    else wait_vmcnt<0>();
    raw_barrier();
    if constexpr (FAST) issue(t + DIST, nxt_off, std::true_type{});
    else if (t + DIST < nt) issue(t + DIST, nxt_off, std::false_type{});
    cur_off = cur_off + BUFB == RINGB ? 0 : cur_off + BUFB;
    nxt_off = nxt_off + BUFB == RINGB ? 0 : nxt_off + BUFB;
    // round 5: a wave all of whose queries lie beyond S (the last query block of a sequence that is not a multiple of 128: S = 264 ->
    // three of the third block's four waves, a quarter of the segment's waves) takes part in the DMA and the barrier and leaves the pipes
    // to the waves that have rows -- it stores nothing, so the results are untouched
    if (!wave_live) return;
    // ---- S^T - m = K (cQ)^T - m : sacc[qt][kt] holds (s2 - mrun)[key = kt*16 + 4g + r][q = li] ----
    f32x4_t sacc[QT][4];
#pragma unroll
    for (int kt = 0; kt < 4; kt++)
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const bf16x8_t kf = RT::frag(k_lds, kt * 16 + li, ks * 4 + g);
#pragma unroll
        for (int qt = 0; qt < QT; qt++)
          sacc[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], ks == 0 ? seed[qt] : sacc[qt][kt], 0, 0, 0);
      }
    if constexpr (!FAST) {
      if (k0 + 64 > S) {  // wave-uniform: only the last tile can hold padded keys
#pragma unroll
        for (int qt = 0; qt < QT; qt++)
#pragma unroll
          for (int kt = 0; kt < 4; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (k0 + kt * 16 + 4 * g + r >= S) sacc[qt][kt][r] = -INFINITY;
      }
    }
    u32x4_t pw[QT][2];
    f32x2_t ls2[QT];
    // exponentials straight from the accumulators, packed to bf16; row-sum partials unless the pad column provides them
    auto exps = [&](const int qt) __attribute__((always_inline)) {
      ls2[qt] = (f32x2_t){0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
          for (int hf = 0; hf < 2; hf++) {
            const int kt = 2 * c + k2;
            const f32x2_t e = {__builtin_amdgcn_exp2f(sacc[qt][kt][2 * hf]), __builtin_amdgcn_exp2f(sacc[qt][kt][2 * hf + 1])};
            if constexpr (RS == 0) ls2[qt] += e;
            pw[qt][c][2 * k2 + hf] = cvt_pk_bf16(e[0], e[1]);
          }
    };
    // ONE wave-uniform decision per tile (all rows of the wave): the common path is a single straight-line block
    bool rebase = FIRST;
    if constexpr (!FIRST) {
      uint32_t orw = 0;
#pragma unroll
      for (int qt = 0; qt < QT; qt++) {
        exps(qt);
        orw |= (pw[qt][0][0] | pw[qt][0][1] | pw[qt][0][2]) | (pw[qt][0][3] | pw[qt][1][0] | pw[qt][1][1]) |
               (pw[qt][1][2] | pw[qt][1][3]);
      }
      rebase = __any((orw & 0x40004000u) != 0);   // some probability of some row of this wave reached 2.0
    }
    if (rebase) {
#pragma unroll
      for (int qt = 0; qt < QT; qt++) {
        // exact row maximum (relative to the current base), new base = old base + max(mx + headroom, 0) -- never lowered
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
#pragma unroll
          for (int r = 0; r < 4; r++) mx = fmaxf(mx, sacc[qt][kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float shift = mx + SM_HEADROOM;
        if constexpr (!FIRST) {
          shift = fmaxf(shift, 0.f);
          const float alpha = __builtin_amdgcn_exp2f(-shift);
          lrun[qt] *= alpha;
          if constexpr (RS == 2) lacc[qt] *= alpha;
#pragma unroll
          for (int dt = 0; dt < DT; dt++) oacc[qt][dt] *= alpha;
        }
        mrun[qt] += shift;
        seed[qt] = (f32x4_t){-mrun[qt], -mrun[qt], -mrun[qt], -mrun[qt]};
#pragma unroll
        for (int kt = 0; kt < 4; kt++) sacc[qt][kt] -= (f32x4_t){shift, shift, shift, shift};
        exps(qt);
      }
    }
    bf16x8_t pf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
      if constexpr (RS == 0) lrun[qt] += ls2[qt][0] + ls2[qt][1];
      pf[qt][0] = __builtin_bit_cast(bf16x8_t, pw[qt][0]);
      pf[qt][1] = __builtin_bit_cast(bf16x8_t, pw[qt][1]);
    }
    // ---- O^T += V^T P^T : oacc[qt][dt] holds O^T[d = dt*16 + 4g + r][q = li] (PSUM: d = hd is the row sum) ----
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const bf16x8_t vf = trf.load(v_lds, c * 32, dt * 16);
#pragma unroll
        for (int qt = 0; qt < QT; qt++)
          oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][c], oacc[qt][dt], 0, 0, 0);
      }
    if constexpr (RS == 2) {
      const u32x4_t one4 = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
      const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, one4);
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int qt = 0; qt < QT; qt++) lacc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[qt][c], lacc[qt], 0, 0, 0);
    }
  };
  const int nfull = RT::CAN_FULL ? S / 64 : 0;
  const int t_fast = nfull - DIST > 0 ? nfull - DIST : 0;
  int t = 1;
  if (t_fast > 0) iter(0, std::true_type{}, std::true_type{});
  else iter(0, std::false_type{}, std::true_type{});
  for (; t < t_fast; t++) iter(t, std::true_type{}, std::false_type{});
  for (; t < nt; t++) iter(t, std::false_type{}, std::false_type{});

#pragma unroll
  for (int qt = 0; qt < QT; qt++) {
    float l;
    if constexpr (PSUM) {
      // column hd = 24 of O^T: tile dt = 1, rows 4g + r = 8 -> lanes g == 2, element 0; broadcast to the row's four lane groups
      l = __shfl(oacc[qt][1][0], 32 + li, 64);
    } else if constexpr (RS == 2) {
      l = lacc[qt][0];   // every row of the block holds the sum of column q = li
    } else {
      l = lrun[qt];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
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
// backward, part 1: dK, dV.  One workgroup per 64*KT-key block (16*KT keys per wave), loop over 64-query tiles.
//   S = Q K^T (lane: S[q = qt*16+4g+r][key = li]),  P = exp2(S*sc - lse2[q]),  dP = dO V^T,
//   dS = P (dP - delta[q]),  dV^T += dO^T P,  dK^T += Q^T dS  (then * scale)
// KT = 16-key tiles per wave.  Every LDS read instruction (ds_read_b128 and the half-as-wide ds_read_b64_tr_b16 alike)
// occupies the CU's LDS pipe for 4 cycles, i.e. costs a SIMD 16 cycles of its share -- as much as one MFMA
// (tools/probes/valu_probe.hip).  With KT = 1 a wave issues 32 (hd <= 32) ... 56 (hd = 64) of them per 64-query tile for
// 16 MFMAs + the soft-max arithmetic of 16 scores per lane: the kernel is bound by LDS INSTRUCTIONS.  The Q / dO fragments
// and their transposed reads do not depend on the key, so KT = 2 reuses every one of them for two key tiles.
// =============================================================================================================
template <int HDP, int KT, bool SM>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(const bf16_t* __restrict__ qkv_all,
                                                            const bf16_t* __restrict__ dout_all,
                                                            const float* __restrict__ lse2_all,
                                                            const float* __restrict__ delta_all,
                                                            bf16_t* __restrict__ dqkv_all, AttnSegs sg, int H, int hd,
                                                            float sc, float scale, float* __restrict__ colkv) {
  // two {Q, dO} images filled by LDS-DMA (tile t+1 lands while tile t is multiplied) + two {lse, delta} rows; one
  // barrier per tile
  constexpr int BUFB = 2 * RowTile<HDP>::BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUFB + 2 * 2 * 64 * 4];
  float* stat = (float*)(smem + 2 * BUFB);   // [buffer][lse 0..63 | -delta 0..63]
  const TrFrag<HDP> trf(threadIdx.x & 63);
  constexpr int KS = HDP / 32, DT = HeadTiles<HDP>::DT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int si = attn_seg_of(sg, logical);
  const int S = sg.S[si], nkb = sg.nb[si];
  const int64_t rs = (int64_t)3 * H * hd;
  const int64_t os = (int64_t)H * hd;
  const bf16_t* qkv = qkv_all + sg.row0[si] * rs;
  const bf16_t* dout = dout_all + sg.row0[si] * os;
  const float* lse2 = lse2_all + (int64_t)H * sg.row0[si];
  const float* delta = delta_all + (int64_t)H * sg.row0[si];
  bf16_t* dqkv = dqkv_all + sg.row0[si] * rs;
  if (colkv != nullptr) colkv += sg.col0[si] * (2 * os);
  const int kb = logical % nkb, bh = logical / nkb;
  const int h = bh % H, b = bh / H;
  const bf16_t* qbase = qkv + (int64_t)b * S * rs + (int64_t)h * hd;
  const bf16_t* kbase = qbase + (int64_t)H * hd;
  const bf16_t* vbase = qbase + (int64_t)2 * H * hd;
  const bf16_t* dobase = dout + (int64_t)b * S * os + (int64_t)h * hd;
  const float* lse_b = lse2 + ((int64_t)b * H + h) * S;
  const float* dl_b = delta + ((int64_t)b * H + h) * S;
  int key[KT];
  bool key_ok[KT];
#pragma unroll
  for (int kt = 0; kt < KT; kt++) {
    key[kt] = kb * (64 * KT) + (w * KT + kt) * 16 + li;
    key_ok[kt] = key[kt] < S;
  }

  bf16x8_t kf[KT][KS], vf[KT][KS];
#pragma unroll
  for (int kt = 0; kt < KT; kt++)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int d0 = ks * 32 + 8 * g;
      kf[kt][ks] = load_frag_global(kbase + (int64_t)key[kt] * rs + d0, key_ok[kt] && d0 < hd);
      vf[kt][ks] = load_frag_global(vbase + (int64_t)key[kt] * rs + d0, key_ok[kt] && d0 < hd);
      // SM: K is the stationary operand of the score product here and is used for nothing else (dK contracts dS with Q):
      // it carries the soft-max scale, bf16(k * scale * log2 e)
      if constexpr (SM) kf[kt][ks] = scale_frag(kf[kt][ks], sc);
    }
  f32x4_t dvacc[KT][DT], dkacc[KT][DT];
#pragma unroll
  for (int kt = 0; kt < KT; kt++)
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
      dvacc[kt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      dkacc[kt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

  float lse_r = 0.f, dl_r = 0.f;
  const f32x2_t sc2 = {sc, sc};
  const int nt = (S + 63) / 64;
  const bool wave_live = kb * (64 * KT) + __builtin_amdgcn_readfirstlane(w) * (16 * KT) < S;   // wave-uniform
  const TileDma<HDP, 256> dma(tid, hd);
  auto load_stats = [&](int r0) {
    if (tid < 64) {
      const int q = r0 + tid;
      lse_r = q < S ? lse_b[q] : INFINITY;  // +inf -> P = 0 for padded query rows (their Q / dO rows re-read row S-1)
      dl_r = q < S ? -dl_b[q] : 0.f;        // NEGATED: it seeds the dP accumulator
    }
  };
  __builtin_amdgcn_s_waitcnt(0x0f70);   // compiler-visible vmcnt(0): the K / V fragment loads are complete
  load_stats(0);
  dma.template issue<false>(qbase, rs, dobase, os, 0, S, smem);
  if (tid < 64) {
    stat[tid] = lse_r;
    stat[64 + tid] = dl_r;
  }
  if (nt > 1) load_stats(64);

  for (int t = 0; t < nt; t++) {
    char* q_lds = smem + (t & 1) * BUFB;
    char* do_lds = q_lds + RowTile<HDP>::BYTES;
    const float* lse_s = stat + (t & 1) * 128;
    const float* dl_s = lse_s + 64;
    // this wave's part of tile t (and the statistics registers of tile t+1) has landed; the barrier publishes every
    // wave's part and the statistics row written an iteration ago, and frees the other buffers
    wait_vmcnt<0>();
    raw_barrier();
    if (t + 1 < nt) {
      if (tid < 64) {
        stat[((t + 1) & 1) * 128 + tid] = lse_r;
        stat[((t + 1) & 1) * 128 + 64 + tid] = dl_r;
      }
      if (t + 2 < nt) load_stats((t + 2) * 64);
      if ((t + 1) * 64 + 64 <= S) dma.template issue<true>(qbase, rs, dobase, os, (t + 1) * 64, S, smem + ((t + 1) & 1) * BUFB);
      else dma.template issue<false>(qbase, rs, dobase, os, (t + 1) * 64, S, smem + ((t + 1) & 1) * BUFB);
    }

    // Two halves of the 64-query tile (c = 0, 1: queries 32c .. 32c+31 = the contraction chunk of one transposed MFMA): S / dP / P / dS of
    // a half, then its dV / dK MFMAs.  Per accumulator the operations and their order are those of the round-4 form (all four
    // 16-query blocks first, then both halves) -- bit-identical -- but only one half's P / dS is live at a time, which is what lets
    // KT = 4 (64 keys per wave: every Q / dO fragment, transposed read and statistics read serves FOUR key tiles) fit the registers.
    // round 5: a wave all of whose keys lie beyond S skips the tile's arithmetic (it stores nothing), and a half all of whose QUERIES are
    // padded rows is skipped by every wave (P = 0 there: it would add zeros to every accumulator)
    if (!wave_live) continue;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      if (c == 1 && t * 64 + 32 >= S) break;
      float pv[KT][2][4], dsv[KT][2][4];
#pragma unroll
      for (int q2 = 0; q2 < 2; q2++) {
        const int qt = 2 * c + q2;
        const float4 lse4 = *(const float4*)(lse_s + qt * 16 + 4 * g);   // queries qt*16 + 4g + {0..3}
        const float4 ndl4 = *(const float4*)(dl_s + qt * 16 + 4 * g);    // -delta of the same queries
        // the dP accumulator STARTS at -delta[q] (this lane's four rows): dP - delta comes out of the matrix pipe
        // SM: the score accumulator starts at -lse2[q] the same way, so P = exp2(accumulator) (padded queries: -inf -> P = 0)
        f32x4_t sacc[KT], dpacc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
          sacc[kt] = SM ? (f32x4_t){-lse4.x, -lse4.y, -lse4.z, -lse4.w} : (f32x4_t){0.f, 0.f, 0.f, 0.f};
          dpacc[kt] = (f32x4_t){ndl4.x, ndl4.y, ndl4.z, ndl4.w};
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
          const bf16x8_t qa = RowTile<HDP>::frag(q_lds, qt * 16 + li, ks * 4 + g);     // read ONCE for all KT key tiles
          const bf16x8_t da = RowTile<HDP>::frag(do_lds, qt * 16 + li, ks * 4 + g);
#pragma unroll
          for (int kt = 0; kt < KT; kt++) {
            sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kt][ks], sacc[kt], 0, 0, 0);
            dpacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kt][ks], dpacc[kt], 0, 0, 0);
          }
        }
        const f32x2_t nl[2] = {{-lse4.x, -lse4.y}, {-lse4.z, -lse4.w}};
#pragma unroll
        for (int kt = 0; kt < KT; kt++)
#pragma unroll
          for (int hf = 0; hf < 2; hf++) {   // 2-vectors: v_pk_fma_f32 / v_pk_mul_f32
            const f32x2_t sv = {sacc[kt][2 * hf], sacc[kt][2 * hf + 1]}, dpv = {dpacc[kt][2 * hf], dpacc[kt][2 * hf + 1]};
            f32x2_t a = sv;
            if constexpr (!SM) a = __builtin_elementwise_fma(sv, sc2, nl[hf]);   // padded queries: lse = +inf -> P = 0
            const f32x2_t e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
            const f32x2_t ds = e * dpv;
            pv[kt][q2][2 * hf] = e[0];
            pv[kt][q2][2 * hf + 1] = e[1];
            dsv[kt][q2][2 * hf] = ds[0];
            dsv[kt][q2][2 * hf + 1] = ds[1];
          }
      }
      bf16x8_t pfr[KT], dfr[KT];
#pragma unroll
      for (int kt = 0; kt < KT; kt++) {
        u32x4_t pw, dw;
        pw[0] = cvt_pk_bf16(pv[kt][0][0], pv[kt][0][1]);
        pw[1] = cvt_pk_bf16(pv[kt][0][2], pv[kt][0][3]);
        pw[2] = cvt_pk_bf16(pv[kt][1][0], pv[kt][1][1]);
        pw[3] = cvt_pk_bf16(pv[kt][1][2], pv[kt][1][3]);
        dw[0] = cvt_pk_bf16(dsv[kt][0][0], dsv[kt][0][1]);
        dw[1] = cvt_pk_bf16(dsv[kt][0][2], dsv[kt][0][3]);
        dw[2] = cvt_pk_bf16(dsv[kt][1][0], dsv[kt][1][1]);
        dw[3] = cvt_pk_bf16(dsv[kt][1][2], dsv[kt][1][3]);
        pfr[kt] = __builtin_bit_cast(bf16x8_t, pw);
        dfr[kt] = __builtin_bit_cast(bf16x8_t, dw);
      }
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const bf16x8_t dot_f = trf.load(do_lds, c * 32, dt * 16);                     // read ONCE for all KT key tiles
        const bf16x8_t qt_f = trf.load(q_lds, c * 32, dt * 16);
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
          dvacc[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_f, pfr[kt], dvacc[kt][dt], 0, 0, 0);
          dkacc[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f, dfr[kt], dkacc[kt][dt], 0, 0, 0);
        }
      }
    }
  }

  // column partials for the qkv bias gradient (colkv != nullptr): this workgroup's sum over its keys of dK | dV (fp32, before the
  // bf16 rounding) -> colkv[b * nkb + kb][h*hd + d | H*hd + h*hd + d]; padded keys (garbage accumulators, never stored) are masked
  if (colkv != nullptr) {
    float ck[DT][4], cv[DT][4];
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; kt++) {
          a += key_ok[kt] ? dkacc[kt][dt][r] : 0.f;
          c += key_ok[kt] ? dvacc[kt][dt][r] : 0.f;
        }
        ck[dt][r] = row16_sum(a);   // over the 16 keys of the lane row (li)
        cv[dt][r] = row16_sum(c);
      }
    raw_barrier();   // every wave is done with the {Q, dO} images: the first 4 x 2 x HDP floats of the ring become the combine buffer
    float* red = (float*)smem;
    if (li == 0) {
#pragma unroll
      for (int dt = 0; dt < DT; dt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          red[(w * 2 + 0) * HDP + dt * 16 + 4 * g + r] = ck[dt][r];
          red[(w * 2 + 1) * HDP + dt * 16 + 4 * g + r] = cv[dt][r];
        }
    }
    raw_barrier();
    if (tid < 2 * HDP) {
      const int which = tid / HDP, d = tid % HDP;
      if (d < hd) {
        float t = red[(0 * 2 + which) * HDP + d] + red[(1 * 2 + which) * HDP + d] + red[(2 * 2 + which) * HDP + d] +
                  red[(3 * 2 + which) * HDP + d];
        if (which == 0) t *= scale;
        colkv[((int64_t)b * nkb + kb) * (2 * os) + (int64_t)which * os + (int64_t)h * hd + d] = t;
      }
    }
  }
#pragma unroll
  for (int kt = 0; kt < KT; kt++) {
    if (key_ok[kt]) {
      bf16_t* dkp = dqkv + ((int64_t)b * S + key[kt]) * rs + (int64_t)H * hd + (int64_t)h * hd;
      bf16_t* dvp = dkp + (int64_t)H * hd;
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const int d = dt * 16 + 4 * g;
        if (d < hd) {
          u32x2_t a, c;
          a[0] = cvt_pk_bf16(dkacc[kt][dt][0] * scale, dkacc[kt][dt][1] * scale);
          a[1] = cvt_pk_bf16(dkacc[kt][dt][2] * scale, dkacc[kt][dt][3] * scale);
          c[0] = cvt_pk_bf16(dvacc[kt][dt][0], dvacc[kt][dt][1]);
          c[1] = cvt_pk_bf16(dvacc[kt][dt][2], dvacc[kt][dt][3]);
          *(u32x2_t*)(dkp + d) = a;
          *(u32x2_t*)(dvp + d) = c;
        }
      }
    }
  }
}

// =============================================================================================================
// backward, part 2: dQ.  One workgroup per 128 queries (32 per wave), loop over 64-key tiles.
//   S^T = K Q^T, dP^T = V dO^T (lane: [key = kt*16+4g+r][q = li]),  dS^T = P^T (dP^T - delta[q]),
//   dQ^T += K^T dS^T  (then * scale)
// =============================================================================================================
//   delta[q] = sum_d dO[q,d] O[q,d] (the softmax-backward row term) is computed HERE, from the dO fragments the wave holds
//   anyway plus one read of its O rows, and written to `delta` for the dK/dV kernel, which is launched after this one:
//   the separate delta pass (one more kernel on the critical path of every attention backward) is gone.
// QW = 16-query tiles per wave (2: 128 queries per workgroup, the form of rounds 2-4; 4: 256 -- every K / V fragment, every transposed
// K read of a key tile then serves four query tiles: the kernel is bound by LDS instruction count at head_dim 24, see the dK/dV kernel)
template <int HDP, bool SM, int QW = 2>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv_all,
                                                          const bf16_t* __restrict__ o_all,
                                                          const bf16_t* __restrict__ dout_all,
                                                          const float* __restrict__ lse2_all,
                                                          float* __restrict__ delta_all,
                                                          bf16_t* __restrict__ dqkv_all, AttnSegs sg, int H, int hd,
                                                          float sc, float scale, float* __restrict__ colq) {
  // {K,V} x NBUF ring filled by LDS-DMA (see the forward kernel): one barrier per tile, tile t+DIST in flight
  constexpr int NBUF = HDP <= 64 ? 3 : 2, DIST = NBUF - 1, BUFB = 2 * RowTile<HDP>::BYTES, RINGB = NBUF * BUFB;
  __shared__ __attribute__((aligned(16))) char smem[RINGB];
  const TrFrag<HDP> trf(threadIdx.x & 63);
  constexpr int KS = HDP / 32, DT = HeadTiles<HDP>::DT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int si = attn_seg_of(sg, logical);
  const int S = sg.S[si], nqb = sg.nb[si];
  const int64_t rs = (int64_t)3 * H * hd;
  const int64_t os = (int64_t)H * hd;
  const bf16_t* qkv = qkv_all + sg.row0[si] * rs;
  const bf16_t* o = o_all + sg.row0[si] * os;
  const bf16_t* dout = dout_all + sg.row0[si] * os;
  const float* lse2 = lse2_all + (int64_t)H * sg.row0[si];
  float* delta = delta_all + (int64_t)H * sg.row0[si];
  bf16_t* dqkv = dqkv_all + sg.row0[si] * rs;
  if (colq != nullptr) colq += sg.col0[si] * os;
  const int qb = logical % nqb, bh = logical / nqb;
  const int h = bh % H, b = bh / H;
  const bf16_t* qbase = qkv + (int64_t)b * S * rs + (int64_t)h * hd;
  const bf16_t* kbase = qbase + (int64_t)H * hd;
  const bf16_t* vbase = qbase + (int64_t)2 * H * hd;
  const bf16_t* dobase = dout + (int64_t)b * S * os + (int64_t)h * hd;
  const bf16_t* obase = o + (int64_t)b * S * os + (int64_t)h * hd;
  const int q0 = qb * (64 * QW) + w * (16 * QW);

  bf16x8_t qf[QW][KS], dof[QW][KS];
  float lse_q[QW], dl_q[QW];
#pragma unroll
  for (int qt = 0; qt < QW; qt++) {
    const int q = q0 + qt * 16 + li;
    float dsum = 0.f;   // this lane's share of delta[q]: head-dim chunks 8g .. 8g+7 of every 32-wide step
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int d0 = ks * 32 + 8 * g;
      const bool ok = q < S && d0 < hd;
      qf[qt][ks] = load_frag_global(qbase + (int64_t)q * rs + d0, ok);
      dof[qt][ks] = load_frag_global(dobase + (int64_t)q * os + d0, ok);
      const u32x4_t ov = __builtin_bit_cast(u32x4_t, load_frag_global(obase + (int64_t)q * os + d0, ok));
      const u32x4_t dv = __builtin_bit_cast(u32x4_t, dof[qt][ks]);
#pragma unroll
      for (int j = 0; j < 4; j++) dsum += bf_lo(ov[j]) * bf_lo(dv[j]) + bf_hi(ov[j]) * bf_hi(dv[j]);
    }
    dsum += __shfl_xor(dsum, 16, 64);   // the four lane groups g hold the four 8-element chunks of a step
    dsum += __shfl_xor(dsum, 32, 64);
    lse_q[qt] = q < S ? lse2[((int64_t)b * H + h) * S + q] : INFINITY;
    dl_q[qt] = q < S ? dsum : 0.f;
    if (g == 0 && q < S) delta[((int64_t)b * H + h) * S + q] = dsum;
    // SM: Q is the stationary operand of the score product and is used for nothing else here: it carries scale * log2 e
    if constexpr (SM) {
#pragma unroll
      for (int ks = 0; ks < KS; ks++) qf[qt][ks] = scale_frag(qf[qt][ks], sc);
    }
  }
  f32x4_t dqacc[QW][DT];
#pragma unroll
  for (int qt = 0; qt < QW; qt++)
#pragma unroll
    for (int dt = 0; dt < DT; dt++) dqacc[qt][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  f32x4_t ndl4[QW];   // -delta[q] of this lane's query, four times: the seed of every dP accumulator block
  f32x4_t nls4[QW];   // SM: -lse2[q] four times: the seed of every score accumulator block (padded queries: -inf -> P = 0)
#pragma unroll
  for (int qt = 0; qt < QW; qt++) {
    ndl4[qt] = (f32x4_t){-dl_q[qt], -dl_q[qt], -dl_q[qt], -dl_q[qt]};
    nls4[qt] = SM ? (f32x4_t){-lse_q[qt], -lse_q[qt], -lse_q[qt], -lse_q[qt]} : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const f32x2_t sc2 = {sc, sc};
  const int nt = (S + 63) / 64;
  const bool wave_live = qb * (64 * QW) + __builtin_amdgcn_readfirstlane(w) * (16 * QW) < S;   // wave-uniform
  const TileDma<HDP, 256> dma(tid, hd);
  constexpr int NDMA2 = 2 * TileDma<HDP, 256>::NDMA;   // DMA instructions per wave per tile
  __builtin_amdgcn_s_waitcnt(0x0f70);   // compiler-visible vmcnt(0): the Q / dO / lse loads above are complete
#pragma unroll
  for (int d = 0; d < DIST; d++)
    if (d < nt) dma.template issue<false>(kbase, rs, vbase, rs, d * 64, S, smem + d * BUFB);
  int cur_off = 0, nxt_off = DIST * BUFB;

  for (int t = 0; t < nt; t++) {
    const int k0 = t * 64;
    char* k_lds = smem + cur_off;
    char* v_lds = k_lds + RowTile<HDP>::BYTES;
    if (DIST >= 2 && t + 1 < nt) wait_vmcnt<NDMA2>();
    else wait_vmcnt<0>();
    raw_barrier();
    if (t + DIST < nt) {
      if ((t + DIST) * 64 + 64 <= S) dma.template issue<true>(kbase, rs, vbase, rs, (t + DIST) * 64, S, smem + nxt_off);
      else dma.template issue<false>(kbase, rs, vbase, rs, (t + DIST) * 64, S, smem + nxt_off);
    }
    cur_off = cur_off + BUFB == RINGB ? 0 : cur_off + BUFB;
    nxt_off = nxt_off + BUFB == RINGB ? 0 : nxt_off + BUFB;
    // Two halves of the 64-key tile (c = 0, 1: keys 32c .. 32c+31 = the contraction chunk of one dQ MFMA): S^T / dP^T / dS^T of a half
    // for all QW query tiles, then its dQ MFMAs.  Per accumulator the same operations in the same order as the round-4 form (all
    // four key blocks first, then both halves): bit-identical; only one half's scores are live, which is what lets QW = 4 fit.
    // round 5: a wave all of whose queries lie beyond S skips the tile's arithmetic (it stores nothing), and a half all of whose KEYS are
    // padded is skipped by every wave (its dS is set to zero below: it would add zeros to every dQ accumulator)
    if (!wave_live) continue;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      if (c == 1 && k0 + 32 >= S) break;
      f32x4_t sacc[QW][2], dpacc[QW][2];
#pragma unroll
      for (int qt = 0; qt < QW; qt++)
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
          sacc[qt][k2] = nls4[qt];    // zeros, or (SM) -lse2[q]: s - lse comes out of the matrix pipe
          dpacc[qt][k2] = ndl4[qt];   // the dP accumulators start at -delta[q]: dP - delta comes out of the matrix pipe
        }
#pragma unroll
      for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
          const int kt = 2 * c + k2;
          const bf16x8_t ka = RowTile<HDP>::frag(k_lds, kt * 16 + li, ks * 4 + g);
          const bf16x8_t va = RowTile<HDP>::frag(v_lds, kt * 16 + li, ks * 4 + g);
#pragma unroll
          for (int qt = 0; qt < QW; qt++) {
            sacc[qt][k2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[qt][ks], sacc[qt][k2], 0, 0, 0);
            dpacc[qt][k2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, dof[qt][ks], dpacc[qt][k2], 0, 0, 0);
          }
        }
      // dS^T = P^T (dP^T - delta)
#pragma unroll
      for (int qt = 0; qt < QW; qt++) {
        const f32x2_t nl = {-lse_q[qt], -lse_q[qt]};
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
          for (int hf = 0; hf < 2; hf++) {   // 2-vectors: v_pk_fma_f32 / v_pk_mul_f32
            const f32x2_t sv = {sacc[qt][k2][2 * hf], sacc[qt][k2][2 * hf + 1]};
            const f32x2_t dpv = {dpacc[qt][k2][2 * hf], dpacc[qt][k2][2 * hf + 1]};
            f32x2_t a = sv;
            if constexpr (!SM) a = __builtin_elementwise_fma(sv, sc2, nl);
            const f32x2_t e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
            const f32x2_t ds = e * dpv;
            sacc[qt][k2][2 * hf] = ds[0];
            sacc[qt][k2][2 * hf + 1] = ds[1];
          }
      }
      if (k0 + 64 > S) {   // wave-uniform, last tile only: padded keys re-read key S-1, their dS must vanish
        asm volatile("" ::: "memory");   // keeps this a real branch (the kernel is VALU-bound; an if-converted mask costs ~30 %)
#pragma unroll
        for (int qt = 0; qt < QW; qt++)
#pragma unroll
          for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (k0 + (2 * c + k2) * 16 + 4 * g + r >= S) sacc[qt][k2][r] = 0.f;
      }
      bf16x8_t dsf[QW];
#pragma unroll
      for (int qt = 0; qt < QW; qt++) {
        u32x4_t dw;
        dw[0] = cvt_pk_bf16(sacc[qt][0][0], sacc[qt][0][1]);
        dw[1] = cvt_pk_bf16(sacc[qt][0][2], sacc[qt][0][3]);
        dw[2] = cvt_pk_bf16(sacc[qt][1][0], sacc[qt][1][1]);
        dw[3] = cvt_pk_bf16(sacc[qt][1][2], sacc[qt][1][3]);
        dsf[qt] = __builtin_bit_cast(bf16x8_t, dw);
      }
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const bf16x8_t ktf = trf.load(k_lds, c * 32, dt * 16);
#pragma unroll
        for (int qt = 0; qt < QW; qt++)
          dqacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qt], dqacc[qt][dt], 0, 0, 0);
      }
    }
  }

  // column partials for the qkv bias gradient (colq != nullptr): this workgroup's sum over its 128 queries of dQ (fp32, before the
  // bf16 rounding) -> colq[b * nqb + qb][h*hd + d]; padded queries have P = 0, hence dQ = 0, and need no mask
  if (colq != nullptr) {
    float cq[DT][4];
#pragma unroll
    for (int dt = 0; dt < DT; dt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float a = dqacc[0][dt][r] + dqacc[1][dt][r];
        if constexpr (QW == 4) a = a + dqacc[2][dt][r] + dqacc[3][dt][r];
        cq[dt][r] = row16_sum(a);
      }
    raw_barrier();   // every wave is done with the {K, V} ring: its first 4 x HDP floats become the combine buffer
    float* red = (float*)smem;
    if (li == 0) {
#pragma unroll
      for (int dt = 0; dt < DT; dt++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[w * HDP + dt * 16 + 4 * g + r] = cq[dt][r];
    }
    raw_barrier();
    if (tid < hd) colq[((int64_t)b * nqb + qb) * os + (int64_t)h * hd + tid] =
        (red[tid] + red[HDP + tid] + red[2 * HDP + tid] + red[3 * HDP + tid]) * scale;
  }
#pragma unroll
  for (int qt = 0; qt < QW; qt++) {
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
static int pick_hdp(int64_t hd) { return hd <= 32 ? 32 : (hd <= 64 ? 64 : (hd <= 80 ? 96 : (hd <= 128 ? 128 : 0))); }
#define LOG2E 1.4426950408889634f

// segment list -> kernel argument (workgroups per segment = B * H * blocks(S)); empty segments are dropped
static int make_segs(const vj_seg_t* segs, int64_t n_segs, int64_t H, int64_t rows_per_block, AttnSegs* out, int64_t* nblk,
                     const char* who) {
  VJ_CHECK_ARG(segs != nullptr && n_segs >= 1 && n_segs <= VJ_ATTN_MAX_SEGS, "%s: 1..%d segments", who, VJ_ATTN_MAX_SEGS);
  int k = 0;
  int64_t tot = 0, col = 0;
  for (int64_t i = 0; i < n_segs; i++) {
    VJ_CHECK_ARG(segs[i].B >= 0 && segs[i].S >= 0 && segs[i].row0 >= 0, "%s: bad segment %ld", who, (long)i);
    if (segs[i].B * segs[i].S == 0) continue;
    const int64_t nb = cdiv64(segs[i].S, rows_per_block);
    tot += segs[i].B * H * nb;
    VJ_CHECK_ARG(tot < (1ll << 31) && segs[i].S < (1ll << 31), "%s: grid too large", who);
    out->blk_end[k] = (int)tot;
    out->S[k] = (int)segs[i].S;
    out->nb[k] = (int)nb;
    out->row0[k] = segs[i].row0;
    out->col0[k] = col;
    col += segs[i].B * nb;
    k++;
  }
  for (int i = k; i < VJ_ATTN_MAX_SEGS; i++) {
    out->blk_end[i] = (int)tot;
    out->S[i] = 0;
    out->nb[i] = 1;
    out->row0[i] = 0;
    out->col0[i] = 0;
  }
  out->n = k;
  *nblk = tot;
  return 0;
}

// One launch over n_segs <= 4 segments of one token-major activation: qkv [M, 3*H*hd], o [M, H*hd], lse2 [H*M] (segment i: rows
// row0_i .. row0_i + B_i*S_i, its lse2 block at H*row0_i laid out [B_i, H, S_i]).
extern "C" int vj_attn_fwd_segs(const void* qkv, void* o, float* lse2, const vj_seg_t* segs, int64_t n_segs, int64_t H,
                                int64_t hd, float scale, hipStream_t stream) {
  VJ_CHECK_ARG(hd % 8 == 0 && pick_hdp(hd) != 0, "vj_attn_fwd: head_dim=%ld unsupported (need %%8==0, <=128)", (long)hd);
  VJ_CHECK_ARG(H > 0, "vj_attn_fwd: bad dims");
  const int64_t rs = 3 * H * hd, os = H * hd;
  AttnSegs sg;
  int64_t nblk = 0;
  if (int rc = make_segs(segs, n_segs, H, 128, &sg, &nblk, "vj_attn_fwd")) return rc;
  if (nblk == 0) return 0;
  // scale < 0: the q part of qkv ALREADY carries |scale| * log2(e) (the qkv GEMM applied it before its bf16 rounding, epilogue 4
  // of vj_gemm_bf16_nt): the kernels' own factor becomes 1 (scale_frag(x, 1) is the identity)
  const float sc = scale < 0.f ? 1.0f : scale * LOG2E;
  // row sums from the matrix pipe: head_dim 24 on the V image's pad column (RS = 1), the other head sizes from an all-ones operand (RS = 2);
  // the vector-add form (RS = 0) of round 3 is no longer instantiated (profiles/r04_abab_rowsums.md)
#define VJ_FWD_SM(HDPV, NB, RSV)                                                                                      \
  hipLaunchKernelGGL((attn_fwd_sm_kernel<HDPV, 2, NB, RSV>), dim3((unsigned)nblk), dim3(256), 0, stream,               \
                     (const bf16_t*)qkv, (bf16_t*)o, lse2, sg, (int)H, (int)hd, sc)
  switch (pick_hdp(hd)) {
    case 32:
      if (hd == 24) VJ_FWD_SM(32, 3, 1);
      else VJ_FWD_SM(32, 3, 2);
      break;
    case 64: VJ_FWD_SM(64, 2, 2); break;
    case 96: VJ_FWD_SM(96, 2, 2); break;
    default: VJ_FWD_SM(128, 2, 2);
  }
#undef VJ_FWD_SM
  VJ_LAUNCH_CHECK("vj_attn_fwd");
  return 0;
}

extern "C" int vj_attn_fwd(const void* qkv, void* o, float* lse2, int64_t B, int64_t S, int64_t H, int64_t hd,
                           float scale, hipStream_t stream) {
  VJ_CHECK_ARG(B >= 0 && S >= 0 && H > 0, "vj_attn_fwd: bad dims");
  const vj_seg_t one = {0, B, S};
  return vj_attn_fwd_segs(qkv, o, lse2, &one, 1, H, hd, scale, stream);
}

extern "C" int64_t vj_attn_bwd_ws_bytes(int64_t B, int64_t S, int64_t H) { return B * S * H * 4; }

// Workspace of vj_attn_bwd / vj_attn_bwd_segs / vj_attn_bwd_colsum for a segment list: delta = rowsum(dO . O), 4 * H bytes per token row
// up to the last row of the list (laid out like lse2).
extern "C" int64_t vj_attn_bwd_segs_ws_bytes(const vj_seg_t* segs, int64_t n_segs, int64_t H, int64_t hd) {
  (void)hd;
  if (segs == nullptr || n_segs <= 0) return 0;
  int64_t rows_end = 0;
  for (int64_t i = 0; i < n_segs; i++)
    if (segs[i].B * segs[i].S > 0 && segs[i].row0 + segs[i].B * segs[i].S > rows_end) rows_end = segs[i].row0 + segs[i].B * segs[i].S;
  return rows_end * H * 4;
}

// dK/dV tiling: 16-key tiles per wave (32 keys at head_dim <= 32: every Q / dO fragment and transposed read serves two key tiles; 64 keys per
// wave and 64 queries per wave in the dQ kernel were options in round 5: -0.09 / +0.23 ms per step, profiles/r05_attn_tiles.md)
static int dkdv_kt(int64_t hd) { return pick_hdp(hd) == 32 ? 2 : 1; }
static int dq_qw(int64_t) { return 2; }
// rows of the column-partial matrices vj_attn_bwd_colsum writes for one [B, S] segment: colq [rows_q][H*hd], colkv [rows_kv][2*H*hd]
extern "C" int vj_attn_bwd_colsum_rows(int64_t B, int64_t S, int64_t hd, int64_t* rows_q, int64_t* rows_kv) {
  VJ_CHECK_ARG(rows_q != nullptr && rows_kv != nullptr && hd % 8 == 0 && pick_hdp(hd) != 0, "vj_attn_bwd_colsum_rows: bad arguments");
  *rows_q = B * cdiv64(S, 64 * dq_qw(hd));
  *rows_kv = B * cdiv64(S, 64 * dkdv_kt(hd));
  return 0;
}

// One launch pair (dQ, then dK/dV) over n_segs <= 4 segments of one token-major activation (layout as vj_attn_fwd_segs; dout / o
// [M, H*hd], dqkv [M, 3*H*hd], ws >= 4*H*M bytes (delta, laid out like lse2)).  colq / colkv (both or neither): column
// partials of dqkv, segment after segment in the order of the list (rows per segment: vj_attn_bwd_colsum_rows).
extern "C" int vj_attn_bwd_segs(const void* qkv, const void* o, const void* dout, const float* lse2, void* dqkv,
                                const vj_seg_t* segs, int64_t n_segs, int64_t H, int64_t hd, float scale, void* ws,
                                int64_t ws_bytes, float* colq, float* colkv, hipStream_t stream) {
  VJ_CHECK_ARG(hd % 8 == 0 && pick_hdp(hd) != 0, "vj_attn_bwd: head_dim=%ld unsupported", (long)hd);
  VJ_CHECK_ARG(H > 0 && (colq == nullptr) == (colkv == nullptr), "vj_attn_bwd: bad arguments");
  const int kt = dkdv_kt(hd);
  AttnSegs sq, sk;
  int64_t gq = 0, gk = 0;
  const int qw = dq_qw(hd);
  if (int rc = make_segs(segs, n_segs, H, 64 * qw, &sq, &gq, "vj_attn_bwd")) return rc;
  if (int rc = make_segs(segs, n_segs, H, 64 * kt, &sk, &gk, "vj_attn_bwd")) return rc;
  if (gq == 0) return 0;
  int64_t rows_end = 0;   // the delta workspace mirrors lse2: H floats per token row up to the last row of the list
  for (int64_t i = 0; i < n_segs; i++)
    if (segs[i].B * segs[i].S > 0 && segs[i].row0 + segs[i].B * segs[i].S > rows_end) rows_end = segs[i].row0 + segs[i].B * segs[i].S;
  VJ_CHECK_ARG(ws != nullptr && ws_bytes >= rows_end * H * 4, "vj_attn_bwd: workspace too small");
  float* delta = (float*)ws;
  // scale < 0: q is stored pre-scaled by c = |scale| * log2(e) (vj_attn_fwd_segs).  The kernels' own score factor is then 1;
  // dQ = |scale| * dS K is unchanged (the gradient of the UNscaled q: what the qkv dgrad / wgrad expect, since the GEMM's
  // column scale is part of the forward map), and dK = |scale| * dS^T Q = (|scale| / c) * dS^T Q' = dS^T Q' / log2(e)
  const bool pre = scale < 0.f;
  const float sabs = fabsf(scale);
  const float sc = pre ? 1.0f : sabs * LOG2E;
  const float kscale = pre ? 1.0f / LOG2E : sabs;
  // dQ first: it also produces delta[b,h,s] = dO . O for the dK/dV kernel behind it on the same stream.
  // dK/dV: KTV 16-key tiles per wave (option attn_dkdv_kt: 0 = per head-dim class, 1 / 2 forced)
#define VJ_BWD_LAUNCH(HDPV, KTV)                                                                                   \
  do {                                                                                                             \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HDPV, true, 2>), dim3((unsigned)gq), dim3(256), 0, stream,               \
                       (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse2, delta, (bf16_t*)dqkv, sq,   \
                       (int)H, (int)hd, sc, sabs, colq);                                                           \
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<HDPV, KTV, true>), dim3((unsigned)gk), dim3(256), 0, stream,           \
                       (const bf16_t*)qkv, (const bf16_t*)dout, lse2, delta, (bf16_t*)dqkv, sk, (int)H, (int)hd,    \
                       sc, kscale, colkv);                                                                         \
  } while (0)
  switch (pick_hdp(hd)) {
    case 32: VJ_BWD_LAUNCH(32, 2); break;
    case 64: VJ_BWD_LAUNCH(64, 1); break;
    case 96: VJ_BWD_LAUNCH(96, 1); break;
    default: VJ_BWD_LAUNCH(128, 1);
  }
#undef VJ_BWD_LAUNCH
  VJ_LAUNCH_CHECK("vj_attn_bwd");
  return 0;
}

extern "C" int vj_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse2, void* dqkv,
                           int64_t B, int64_t S, int64_t H, int64_t hd, float scale, void* ws, int64_t ws_bytes,
                           hipStream_t stream) {
  VJ_CHECK_ARG(B >= 0 && S >= 0, "vj_attn_bwd: bad dims");
  const vj_seg_t one = {0, B, S};
  return vj_attn_bwd_segs(qkv, o, dout, lse2, dqkv, &one, 1, H, hd, scale, ws, ws_bytes, nullptr, nullptr, stream);
}

// vj_attn_bwd + the column sums of dqkv over this segment's tokens as fp32 partials (the qkv bias gradient, autograd of
// Attention.qkv's bias, modules.py:63): colq[rows_q][H*hd] from the dQ kernel (one row per (sample, 128-query block)), colkv
// [rows_kv][2*H*hd] from the dK/dV kernel (one row per (sample, key block)); every element of both is written.  dqkv is
// bit-identical to vj_attn_bwd's.  Reduce with vj_reduce_segments.
extern "C" int vj_attn_bwd_colsum(const void* qkv, const void* o, const void* dout, const float* lse2, void* dqkv,
                                  int64_t B, int64_t S, int64_t H, int64_t hd, float scale, void* ws, int64_t ws_bytes,
                                  float* colq, float* colkv, hipStream_t stream) {
  VJ_CHECK_ARG(colq != nullptr && colkv != nullptr && B >= 0 && S >= 0, "vj_attn_bwd_colsum: bad arguments");
  const vj_seg_t one = {0, B, S};
  return vj_attn_bwd_segs(qkv, o, dout, lse2, dqkv, &one, 1, H, hd, scale, ws, ws_bytes, colq, colkv, stream);
}
