// Cross-attention of a FEW learned query tokens against the token sequence of a frozen encoder, forward and backward,
// for gfx950: the attention inside the attentive probe that the reference trains on frozen V-JEPA features.
//   reference: CrossAttention.forward (src/models/utils/modules.py:140-157: q = Linear(query), kv = Linear(x) packed
//   [B,N,2,H,hd], F.scaled_dot_product_attention(q, k, v), default scale hd^-0.5; NOTE its `proj` is never applied) as used by
//   CrossAttentionBlock (modules.py:177-181: q + xattn(q, norm1(x))) inside AttentivePooler (attentive_pooler.py:96-102,
//   num_queries = 1 in AttentiveClassifier, attentive_pooler.py:120-130).
//
// One query row against N keys is a matrix-VECTOR product: 2 N hd flop for 4 N hd bytes of K and V -- HBM-bound by two
// orders of magnitude, so there is no MFMA here.  One workgroup per (batch, head[, query]) streams K then V once each
// (16-byte row chunks, eight consecutive lanes cover one 128-byte row of hd = 64), keeps the N scores / probabilities in LDS
// (fp32) and reduces in a fixed order (deterministic).  Algorithmic bytes: forward 4 N hd per (b, h, query); backward
// reads K and V twice and writes dK and dV once: 12 N hd.
//   forward : s_j = (q . k_j) scale log2e ; p_j = 2^(s_j - max) ; y = sum_j p_j v_j / sum_j p_j ; out = residual + y ;
//             lse2 = max + log2(sum)
//   backward: p_j = 2^(s_j - lse2) ; dP_j = dy . v_j ; delta = sum_j p_j dP_j ; dS_j = p_j (dP_j - delta) ;
//             dq = scale sum_j dS_j k_j ; dk_j = scale dS_j q ; dv_j = p_j dy            (one query per (b, h))
#include "common.hpp"

#define XA_THREADS 256
#define XA_LOG2E 1.4426950408889634f

namespace {

__device__ __forceinline__ float xa_block_reduce(float v, float* red, bool is_max) {
  // wave reduction, then the four waves through LDS in a fixed order
  v = is_max ? wave_max(v) : wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();   // `red` may still be read from a previous call
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < XA_THREADS / 64; i++) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  return r;
}

// s_j for every key of this (b, h) into LDS; qv = this head's query row (fp32, LDS), pre-multiplied by scale * log2e
__device__ __forceinline__ void xa_scores(const bf16_t* __restrict__ kbase, int64_t rs, int N, int hd, const float* qv,
                                          float* sc) {
  const int nch = hd >> 3;
  for (int j = threadIdx.x; j < N; j += XA_THREADS) {
    const bf16_t* kp = kbase + (int64_t)j * rs;
    float a = 0.f;
    for (int c = 0; c < nch; c++) {
      const u32x4_t w = *(const u32x4_t*)(kp + c * 8);
#pragma unroll
      for (int i = 0; i < 4; i++) a += bf_lo(w[i]) * qv[c * 8 + 2 * i] + bf_hi(w[i]) * qv[c * 8 + 2 * i + 1];
    }
    sc[j] = a;
  }
}

// out[d] = sum_j wgt[j] * rows[j][d] for this head: thread = (row group rg, 8-column chunk c); partial sums through LDS
// (part[rg][hd], fixed summation order).  Returns the total for column d = threadIdx.x (valid for threadIdx.x < hd).
__device__ __forceinline__ float xa_weighted_rows(const bf16_t* __restrict__ base, int64_t rs, int N, int hd,
                                                  const float* wgt, float* part) {
  const int nch = hd >> 3, ngrp = XA_THREADS / nch;   // nch in {1..16}: 256 / nch row groups (hd = 80: 25 groups, 6 idle threads)
  const int c = threadIdx.x % nch, rg = threadIdx.x / nch;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rg < ngrp) {
    for (int j = rg; j < N; j += ngrp) {
      const u32x4_t w = *(const u32x4_t*)(base + (int64_t)j * rs + c * 8);
      const float p = wgt[j];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        acc[2 * i] += p * bf_lo(w[i]);
        acc[2 * i + 1] += p * bf_hi(w[i]);
      }
    }
  }
  __syncthreads();   // `part` may still be read from a previous call
  if (rg < ngrp) {
#pragma unroll
    for (int i = 0; i < 8; i++) part[rg * hd + c * 8 + i] = acc[i];
  }
  __syncthreads();
  float tot = 0.f;
  if ((int)threadIdx.x < hd)
    for (int g = 0; g < ngrp; g++) tot += part[g * hd + threadIdx.x];
  return tot;
}

// dynamic LDS layout: [qv: 128][red: 8][part: (256 / nch) * hd <= 256 * 8 = 2048][sc: N][ds: N (backward only)]
#define XA_LDS_FIXED (128 + 8 + 2048)

__global__ __launch_bounds__(XA_THREADS) void xattn_fwd_kernel(const bf16_t* __restrict__ q, int64_t q_bstride,
                                                               const bf16_t* __restrict__ kv,
                                                               const bf16_t* __restrict__ resid, bf16_t* __restrict__ out,
                                                               float* __restrict__ lse2, int B, int NQ, int N, int H,
                                                               int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  float* qv = xs;
  float* red = xs + 128;
  float* part = xs + 136;
  float* sc = xs + XA_LDS_FIXED;
  const int iq = blockIdx.x % NQ, bh = blockIdx.x / NQ;
  const int h = bh % H, b = bh / H;
  const int64_t D = (int64_t)H * hd, rs = 2 * D;
  const bf16_t* kbase = kv + (int64_t)b * N * rs + (int64_t)h * hd;
  const bf16_t* vbase = kbase + D;
  const bf16_t* qp = q + (int64_t)b * q_bstride + (int64_t)iq * D + (int64_t)h * hd;
  if ((int)threadIdx.x < hd) qv[threadIdx.x] = bf2f(qp[threadIdx.x]) * (scale * XA_LOG2E);
  __syncthreads();
  xa_scores(kbase, rs, N, hd, qv, sc);
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < N; j += XA_THREADS) mx = fmaxf(mx, sc[j]);   // own entries: no barrier needed yet
  mx = xa_block_reduce(mx, red, true);
  float sum = 0.f;
  for (int j = threadIdx.x; j < N; j += XA_THREADS) {
    const float p = __builtin_amdgcn_exp2f(sc[j] - mx);
    sc[j] = p;
    sum += p;
  }
  sum = xa_block_reduce(sum, red, false);   // its barriers also publish every thread's p_j
  const float y = xa_weighted_rows(vbase, rs, N, hd, sc, part);
  if ((int)threadIdx.x < hd) {
    const int64_t o = ((int64_t)b * NQ + iq) * D + (int64_t)h * hd + threadIdx.x;
    float v = y / sum;
    // the block's residual (the un-projected query tokens, attentive_pooler.py:97-98 + modules.py:178-179) is
    // the same row for every sample
    if (resid) v += bf2f(resid[(int64_t)iq * D + (int64_t)h * hd + threadIdx.x]);
    out[o] = f2bf(v);
  }
  if (threadIdx.x == 0 && lse2) lse2[((int64_t)b * H + h) * NQ + iq] = mx + log2f(sum);
}

__global__ __launch_bounds__(XA_THREADS) void xattn_bwd_kernel(const bf16_t* __restrict__ q, int64_t q_bstride,
                                                               const bf16_t* __restrict__ kv,
                                                               const bf16_t* __restrict__ dy,
                                                               const float* __restrict__ lse2, bf16_t* __restrict__ dq,
                                                               bf16_t* __restrict__ dkv, int B, int N, int H, int hd,
                                                               float scale) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  float* qv = xs;            // q * scale * log2e
  float* red = xs + 128;
  float* part = xs + 136;
  float* sc = xs + XA_LDS_FIXED;   // p_j
  float* ds = sc + N;              // dP_j, then dS_j
  __shared__ float dyv[128], qraw[128];
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const int64_t D = (int64_t)H * hd, rs = 2 * D;
  const bf16_t* kbase = kv + (int64_t)b * N * rs + (int64_t)h * hd;
  const bf16_t* vbase = kbase + D;
  const bf16_t* qp = q + (int64_t)b * q_bstride + (int64_t)h * hd;
  if ((int)threadIdx.x < hd) {
    const float qf = bf2f(qp[threadIdx.x]);
    qraw[threadIdx.x] = qf;
    qv[threadIdx.x] = qf * (scale * XA_LOG2E);
    dyv[threadIdx.x] = bf2f(dy[(int64_t)b * D + (int64_t)h * hd + threadIdx.x]);
  }
  __syncthreads();
  xa_scores(kbase, rs, N, hd, qv, sc);    // s_j
  xa_scores(vbase, rs, N, hd, dyv, ds);   // dP_j = dy . v_j (same row-dot routine, dy as the vector)
  const float l2 = lse2[(int64_t)b * H + h];
  float dl = 0.f;
  for (int j = threadIdx.x; j < N; j += XA_THREADS) {
    const float p = __builtin_amdgcn_exp2f(sc[j] - l2);
    sc[j] = p;
    dl += p * ds[j];
  }
  dl = xa_block_reduce(dl, red, false);   // delta = sum_j p_j dP_j
  for (int j = threadIdx.x; j < N; j += XA_THREADS) ds[j] = sc[j] * (ds[j] - dl);
  __syncthreads();
  // dq = scale * sum_j dS_j k_j   (per sample; the caller sums it over the batch when the projected query is shared)
  const float dqv = xa_weighted_rows(kbase, rs, N, hd, ds, part);
  if ((int)threadIdx.x < hd) dq[(int64_t)b * D + (int64_t)h * hd + threadIdx.x] = f2bf(dqv * scale);
  // dk_j = scale dS_j q ; dv_j = p_j dy : thread = (row group, 8-column chunk), 16-byte stores
  const int nch = hd >> 3, ngrp = XA_THREADS / nch;
  const int c = threadIdx.x % nch, rg = threadIdx.x / nch;
  if (rg < ngrp) {
    float q8[8], d8[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      q8[i] = qraw[c * 8 + i] * scale;
      d8[i] = dyv[c * 8 + i];
    }
    bf16_t* dkb = dkv + (int64_t)b * N * rs + (int64_t)h * hd + c * 8;
    for (int j = rg; j < N; j += ngrp) {
      const float s = ds[j], p = sc[j];
      u32x4_t wk, wv;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        wk[i] = pack_bf2(s * q8[2 * i], s * q8[2 * i + 1]);
        wv[i] = pack_bf2(p * d8[2 * i], p * d8[2 * i + 1]);
      }
      *(u32x4_t*)(dkb + (int64_t)j * rs) = wk;
      *(u32x4_t*)(dkb + (int64_t)j * rs + D) = wv;
    }
  }
}

int xa_check(const char* who, int64_t B, int64_t NQ, int64_t N, int64_t H, int64_t hd, int64_t* lds_bytes, int arrays) {
  VJ_CHECK_ARG(B >= 0 && NQ >= 1 && N >= 1 && H >= 1, "%s: bad dims", who);
  VJ_CHECK_ARG(hd % 8 == 0 && hd >= 8 && hd <= 128, "%s: head_dim=%ld unsupported (need %%8==0, <=128)", who, (long)hd);
  VJ_CHECK_ARG(B * H * NQ < (1ll << 31), "%s: grid too large", who);
  *lds_bytes = (XA_LDS_FIXED + arrays * N) * 4;
  VJ_CHECK_ARG(*lds_bytes <= 160 * 1024 - 2048, "%s: N=%ld keys do not fit the LDS score buffer (max %ld)", who, (long)N,
               (long)((160 * 1024 - 2048) / 4 - XA_LDS_FIXED) / arrays);
  return 0;
}

}  // namespace

extern "C" int vj_xattn_fwd(const void* q, int64_t q_bstride, const void* kv, const void* resid, void* out, float* lse2,
                            int64_t B, int64_t NQ, int64_t N, int64_t H, int64_t hd, float scale, hipStream_t stream) {
  int64_t lds = 0;
  if (int rc = xa_check("vj_xattn_fwd", B, NQ, N, H, hd, &lds, 1)) return rc;
  if (B == 0) return 0;
  static VjPerDeviceOnce once;
  once([] { (void)hipFuncSetAttribute((const void*)xattn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048); });
  hipLaunchKernelGGL(xattn_fwd_kernel, dim3((unsigned)(B * H * NQ)), dim3(XA_THREADS), (size_t)lds, stream, (const bf16_t*)q,
                     q_bstride, (const bf16_t*)kv, (const bf16_t*)resid, (bf16_t*)out, lse2, (int)B, (int)NQ, (int)N, (int)H,
                     (int)hd, scale);
  VJ_LAUNCH_CHECK("vj_xattn_fwd");
  return 0;
}

extern "C" int vj_xattn_bwd(const void* q, int64_t q_bstride, const void* kv, const void* dy, const float* lse2, void* dq,
                            void* dkv, int64_t B, int64_t NQ, int64_t N, int64_t H, int64_t hd, float scale,
                            hipStream_t stream) {
  VJ_CHECK_ARG(NQ == 1, "vj_xattn_bwd: one query per sample (AttentiveClassifier, attentive_pooler.py:120); got %ld", (long)NQ);
  int64_t lds = 0;
  if (int rc = xa_check("vj_xattn_bwd", B, NQ, N, H, hd, &lds, 2)) return rc;
  if (B == 0) return 0;
  static VjPerDeviceOnce once;
  once([] { (void)hipFuncSetAttribute((const void*)xattn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048); });
  hipLaunchKernelGGL(xattn_bwd_kernel, dim3((unsigned)(B * H)), dim3(XA_THREADS), (size_t)lds, stream, (const bf16_t*)q, q_bstride,
                     (const bf16_t*)kv, (const bf16_t*)dy, lse2, (bf16_t*)dq, (bf16_t*)dkv, (int)B, (int)N, (int)H, (int)hd, scale);
  VJ_LAUNCH_CHECK("vj_xattn_bwd");
  return 0;
}
