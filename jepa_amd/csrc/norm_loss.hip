// LayerNorm forward/backward (row per wave, fp32 statistics), the fused target path
// h = F.layer_norm(norm(x))[masks_pred] and the latent L_p loss of the V-JEPA step.
//
// Reference behaviour restated (never copied):
//   nn.LayerNorm(eps=1e-6) in every Block / final norm   src/models/vision_transformer.py:252-281, modules.py:97,106
//   h = F.layer_norm(target_encoder(c)) ; apply_masks    app/vjepa/train.py:424-428
//   loss_fn: mean(|z-h|^p)/p averaged over masks         app/vjepa/train.py:440-446
//   reg_fn : sqrt(var_tokens(z)+1e-4)                    app/vjepa/train.py:448-449,458
#include "common.hpp"
#include "options.hpp"

int vj_reduce_partials_multi(const float* part, float* const* outs, int nseg, int64_t P, int64_t D, float alpha, float beta,
                             hipStream_t stream);   // rows.hip
int vj_reduce_partials_strided(const float* part, float* out, int64_t P, int64_t N, int64_t stride, float alpha,
                               float beta, hipStream_t stream);

#define LN_MAX_CHUNKS 4  // D <= 2048, D % 8 == 0: each lane owns up to 4 chunks of 8 columns

__device__ __forceinline__ void load8(const bf16_t* p, float* v) {
  const u32x4_t w = *(const u32x4_t*)p;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    v[2 * i] = bf_lo(w[i]);
    v[2 * i + 1] = bf_hi(w[i]);
  }
}
__device__ __forceinline__ void store8(bf16_t* p, const float* v) {
  u32x4_t w;
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
  *(u32x4_t*)p = w;
}

// ---------------------------------------------------------------------------------------------
// layernorm_fwd: y = (x-mean)*rstd*gamma + beta  (bf16 in, bf16 out, fp32 math, biased variance)
// ---------------------------------------------------------------------------------------------
// gamma / beta are staged once per workgroup in LDS (fp32, in the order the lanes consume them: every lane's two
// float4 halves of a chunk sit 1 KB apart, so a wave's ds_read_b128 covers 1 KB contiguously): re-reading them from
// global memory for every row put 4x more bytes through the vector cache than the rows themselves (8 KB of affine
// parameters against 2 KB of x per row at D = 1024).
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int64_t rows, int D, float eps) {
  __shared__ __attribute__((aligned(16))) float4 gbs[LN_MAX_CHUNKS][4][64];   // [chunk][gamma lo, gamma hi, beta lo, beta hi][lane]
  const int lane = threadIdx.x & 63;
  for (int q = threadIdx.x; q < LN_MAX_CHUNKS * 64; q += 256) {
    const int i = q >> 6, l = q & 63, c = l * 8 + i * 512;
    if (c < D) {
      gbs[i][0][l] = *(const float4*)(gamma + c);
      gbs[i][1][l] = *(const float4*)(gamma + c + 4);
      gbs[i][2][l] = *(const float4*)(beta + c);
      gbs[i][3][l] = *(const float4*)(beta + c + 4);
    }
  }
  __syncthreads();
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const float invD = 1.0f / (float)D;
  for (int64_t r = wave; r < rows; r += nw) {
    const bf16_t* xp = x + r * D;
    float v[LN_MAX_CHUNKS][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
        load8(xp + c, v[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) s += v[i][j];
      }
    }
    const float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float d = v[i][j] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * invD + eps);
    bf16_t* yp = y + r * D;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
        const float4 g0 = gbs[i][0][lane], g1 = gbs[i][1][lane], b0 = gbs[i][2][lane], b1 = gbs[i][3][lane];
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (v[i][j] - mean) * rstd * gm[j] + bt[j];
        store8(yp + c, o);
      }
    }
    if (lane == 0 && mean_out) {
      mean_out[r] = mean;
      rstd_out[r] = rstd;
    }
  }
}

static inline int ln_grid(int64_t rows) {
  int64_t g = cdiv64(rows, 4);
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int vj_layernorm_fwd(const void* x_bf16, const float* gamma, const float* beta, void* y_bf16, float* mean,
                                float* rstd, int64_t rows, int64_t D, float eps, hipStream_t stream) {
  VJ_CHECK_ARG(D % 8 == 0 && D <= 512 * LN_MAX_CHUNKS, "vj_layernorm_fwd: D=%ld unsupported (need D%%8==0, D<=%d)",
               (long)D, 512 * LN_MAX_CHUNKS);
  VJ_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "vj_layernorm_fwd: mean and rstd must both be given or both null");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ln_grid(rows)), dim3(256), 0, stream, (const bf16_t*)x_bf16, gamma,
                     beta, (bf16_t*)y_bf16, mean, rstd, rows, (int)D, eps);
  VJ_LAUNCH_CHECK("vj_layernorm_fwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm folded into the consuming Linear (round 5; vj_gemm_bf16_nt_lnfold, gemm.hip): what is left of the LayerNorm pass.
//   rowstats: rs[m] = {rstd_m, -mean_m * rstd_m} of the bf16 rows x -- the same two-pass fp32 statistics as layernorm_fwd_kernel
//             (mean, then the variance of the centred values: no E[x^2] - mean^2 cancellation), but x is only READ: half the bytes
//             of the LayerNorm pass, and the GEMM that consumed its output now reads x directly.
//   fold_weights: Wf[n,:] = bf16(W[n,:] * gamma), c[n] = sum_k Wf[n,k] (of the ROUNDED values: the epilogue's acc - mean * c then
//             cancels exactly what the matrix pipe accumulated), bf[n] = b[n] + sum_k W[n,k] beta[k].  One wave per output row;
//             run once per optimizer step on the EMA target's fp32 weights.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_rowstats_kernel(const bf16_t* __restrict__ x, float* __restrict__ rs, int64_t rows, int D,
                                                          float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const float invD = 1.0f / (float)D;
  for (int64_t r = wave; r < rows; r += nw) {
    const bf16_t* xp = x + r * D;
    float v[LN_MAX_CHUNKS][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
        load8(xp + c, v[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) s += v[i][j];
      }
    }
    const float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float d = v[i][j] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * invD + eps);
    if (lane == 0) *(float2*)(rs + 2 * r) = make_float2(rstd, -mean * rstd);
  }
}

extern "C" int vj_ln_rowstats(const void* x_bf16, float* rowstats, int64_t rows, int64_t D, float eps, hipStream_t stream) {
  VJ_CHECK_ARG(D % 8 == 0 && D <= 512 * LN_MAX_CHUNKS, "vj_ln_rowstats: D=%ld unsupported (need D%%8==0, D<=%d)", (long)D, 512 * LN_MAX_CHUNKS);
  VJ_CHECK_ARG(rowstats != nullptr && ((uintptr_t)rowstats % 8 == 0), "vj_ln_rowstats: rowstats null or misaligned");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(ln_rowstats_kernel, dim3(ln_grid(rows)), dim3(256), 0, stream, (const bf16_t*)x_bf16, rowstats, rows, (int)D, eps);
  VJ_LAUNCH_CHECK("vj_ln_rowstats");
  return 0;
}

__global__ __launch_bounds__(256) void ln_fold_weights_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              bf16_t* __restrict__ Wf, float* __restrict__ cvec, float* __restrict__ bf,
                                                              int64_t N, int K) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* wp = W + n * K;
  bf16_t* op = Wf + n * K;
  float sc = 0.f, sb = 0.f;
  for (int k = lane * 4; k < K; k += 256) {   // K % 4 == 0
    const float4 w = *(const float4*)(wp + k);
    const float4 g = *(const float4*)(gamma + k);
    const float4 be = *(const float4*)(beta + k);
    u32x2_t o;
    o[0] = pack_bf2(w.x * g.x, w.y * g.y);
    o[1] = pack_bf2(w.z * g.z, w.w * g.w);
    *(u32x2_t*)(op + k) = o;
    sc += (bf_lo(o[0]) + bf_hi(o[0])) + (bf_lo(o[1]) + bf_hi(o[1]));
    sb += (w.x * be.x + w.y * be.y) + (w.z * be.z + w.w * be.w);
  }
  sc = wave_sum(sc);
  sb = wave_sum(sb);
  if (lane == 0) {
    cvec[n] = sc;
    bf[n] = (b ? b[n] : 0.f) + sb;
  }
}

// W [N,K] fp32 (row-major, contiguous), b [N] fp32 or null, gamma / beta [K] fp32 -> Wf [N,K] bf16, cvec [N], bf [N]
extern "C" int vj_ln_fold_weights(const float* W, const float* b, const float* gamma, const float* beta, void* Wf_bf16, float* cvec,
                                  float* bf, int64_t N, int64_t K, hipStream_t stream) {
  VJ_CHECK_ARG(W != nullptr && gamma != nullptr && beta != nullptr && Wf_bf16 != nullptr && cvec != nullptr && bf != nullptr,
               "vj_ln_fold_weights: null pointer");
  VJ_CHECK_ARG(K % 4 == 0 && K > 0 && N >= 0 && K < (1 << 30), "vj_ln_fold_weights: K=%ld must be a positive multiple of 4", (long)K);
  VJ_CHECK_ARG((((uintptr_t)W | (uintptr_t)gamma | (uintptr_t)beta) % 16 == 0) && ((uintptr_t)Wf_bf16 % 8 == 0),
               "vj_ln_fold_weights: W / gamma / beta must be 16-byte aligned, Wf 8-byte aligned");
  if (N == 0) return 0;
  hipLaunchKernelGGL(ln_fold_weights_kernel, dim3((unsigned)cdiv64(N, 4)), dim3(256), 0, stream, W, b, gamma, beta, (bf16_t*)Wf_bf16,
                     cvec, bf, N, (int)K);
  VJ_LAUNCH_CHECK("vj_ln_fold_weights");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// layernorm_bwd: dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) [+ dres], g = dy*gamma
// per-block partial dgamma/dbeta in fp32 -> part[blk][0:D]=dgamma, part[blk][D:2D]=dbeta
// CS: also the column sums of the OUTPUT dx (fp32, before the bf16 rounding) -> part[blk][2D:3D].  In a transformer block
// dx of norm2's backward is the dY of the proj Linear and dx of norm1's backward is the dY of the previous block's fc2, so
// their bias gradients (colsum of dY) come out of this pass for 16 more accumulator registers instead of costing a
// separate read of dY each (the transpose-free weight-gradient route has no transpose pass to fold them into).
// ---------------------------------------------------------------------------------------------
#define LN_BWD_MAX_BLOCKS 1024
template <int NCH, bool CS, bool PF = true>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in,
                                                            const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                            float* __restrict__ part, int64_t rows, int D) {
  constexpr int NSEG = CS ? 3 : 2;
  __shared__ float red[4][512 * NSEG];  // 4 waves x (512 dgamma | 512 dbeta [| 512 colsum dx]) staged per chunk pass
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float invD = 1.0f / (float)D;
  float ag[NCH][8], ab[NCH][8], gam[NCH][8], as[CS ? NCH : 1][8];
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    const int c = lane * 8 + i * 512;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      ag[i][j] = ab[i][j] = 0.f;
      if constexpr (CS) as[i][j] = 0.f;
      gam[i][j] = (c < D) ? gamma[c + j] : 0.f;
    }
  }
  const int64_t rows_per = cdiv64(rows, gridDim.x);
  const int64_t rbeg = (int64_t)blockIdx.x * rows_per;
  const int64_t rend = (rbeg + rows_per < rows) ? rbeg + rows_per : rows;
  // Software prefetch, one row ahead (round 4): a wave used to load x | dy of its row, wait, reduce, THEN load the residual gradient, and
  // read the row's mean / rstd with a dependent scalar-sized load at the top of every iteration -- three exposed memory latencies per
  // row with ~10 waves per CU on the context-encoder launches (660 workgroups): 3.2 TB/s.  Now the raw 16-byte chunks of x, dy, dres
  // and the two statistics of row r + 4 are requested before row r is computed.  Same arithmetic in the same order: bit-identical.
  u32x4_t nx[NCH], nd[NCH], nr[NCH];
  float nmean = 0.f, nrstd = 0.f;
  auto request = [&](int64_t r) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
        nx[i] = *(const u32x4_t*)(x + r * D + c);
        nd[i] = *(const u32x4_t*)(dy + r * D + c);
        if (dres) nr[i] = *(const u32x4_t*)(dres + r * D + c);
      }
    }
    nmean = mean_in[r];
    nrstd = rstd_in[r];
  };
  auto unpack8 = [](const u32x4_t& w, float* v) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      v[2 * i] = bf_lo(w[i]);
      v[2 * i + 1] = bf_hi(w[i]);
    }
  };
#pragma unroll
  for (int i = 0; i < NCH; i++) nx[i] = nd[i] = nr[i] = (u32x4_t){0u, 0u, 0u, 0u};
  if (PF && rbeg + wv < rend) request(rbeg + wv);
  for (int64_t r = rbeg + wv; r < rend; r += 4) {
    if constexpr (!PF) request(r);   // option ln_bwd_prefetch = 0 (A/B): the row is requested when it is needed
    u32x4_t cx[NCH], cd[NCH], cr[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      cx[i] = nx[i];
      cd[i] = nd[i];
      cr[i] = nr[i];
    }
    const float mean = nmean, rstd = nrstd;
    if (PF && r + 4 < rend) request(r + 4);
    // (the second pass recomputes xhat and g from the raw chunks -- the same operations on the same operands, hence the same bits --
    //  instead of keeping 16 floats per chunk alive across the two wave reductions: with the prefetch buffers that keeps the
    //  D = 1024 variant at three waves per SIMD, i.e. its 660 workgroups resident in one round)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
        float xv[8], dv[8];
        unpack8(cx[i], xv);
        unpack8(cd[i], dv);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float xh = __fmul_rn(__fsub_rn(xv[j], mean), rstd);   // (rounded products: never contracted into the FMAs below,
          const float g = __fmul_rn(dv[j], gam[i][j]);                //  so both passes see the same xhat and g)
          s1 += g;
          s2 += g * xh;
          ag[i][j] += dv[j] * xh;
          ab[i][j] += dv[j];
        }
      }
    }
    const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
        float o[8], xv[8], dv[8];
        unpack8(cx[i], xv);
        unpack8(cd[i], dv);
        if (dres) {
          unpack8(cr[i], o);
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) o[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float xh = __fmul_rn(__fsub_rn(xv[j], mean), rstd);   // (rounded products: never contracted into the FMAs below,
          const float g = __fmul_rn(dv[j], gam[i][j]);                //  so both passes see the same xhat and g)
          o[j] += rstd * (g - c1 - xh * c2);
        }
        if constexpr (CS) {
#pragma unroll
          for (int j = 0; j < 8; j++) as[i][j] += o[j];
        }
        store8(dx + r * D + c, o);
      }
    }
  }
  // cross-wave reduction of the column partials, one chunk pass at a time (512 columns x {dgamma,dbeta})
  float* pg = part + (int64_t)blockIdx.x * NSEG * D;
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    if (i * 512 < D) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; j++) {
        red[wv][lane * 8 + j] = ag[i][j];
        red[wv][512 + lane * 8 + j] = ab[i][j];
        if constexpr (CS) red[wv][1024 + lane * 8 + j] = as[i][j];
      }
      __syncthreads();
      for (int q = threadIdx.x; q < 512 * NSEG; q += 256) {
        const int col = (q & 511) + i * 512;
        if (col < D) {
          const float s = red[0][q] + red[1][q] + red[2][q] + red[3][q];
          pg[(q >> 9) * D + col] = s;
        }
      }
    }
  }
}

extern "C" int64_t vj_layernorm_bwd_ws_bytes(int64_t D) { return (int64_t)LN_BWD_MAX_BLOCKS * 3 * D * 4; }

// The backward kernel alone: dx is complete, the column partials part[nb][nseg * D] (nseg = 3 with `cs`: dgamma | dbeta |
// column sums of dx; else 2) are left in `ws` for the caller to reduce (the block chain reduces everything a block produced
// in ONE launch, chain.hip).  *nb_out = number of partial rows.
int vj_layernorm_bwd_partials(const void* dy_bf16, const void* x_bf16, const float* gamma, const float* mean, const float* rstd,
                              const void* dres_bf16, void* dx_bf16, bool cs, int64_t rows, int64_t D, void* ws,
                              int64_t ws_bytes, int64_t* nb_out, hipStream_t stream) {
  VJ_CHECK_ARG(D % 8 == 0 && D <= 512 * LN_MAX_CHUNKS, "vj_layernorm_bwd: D=%ld unsupported", (long)D);
  VJ_CHECK_ARG(ws_bytes >= vj_layernorm_bwd_ws_bytes(D), "vj_layernorm_bwd: workspace too small");
  *nb_out = 0;
  if (rows == 0) return 0;
  int64_t nb = cdiv64(rows, 16);  // >= 16 rows per workgroup so the column partials amortise
  if (nb > LN_BWD_MAX_BLOCKS) nb = LN_BWD_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  // (PF = true: the row-ahead prefetch of round 4; its control without the prefetch is no longer instantiated: bit-identical, profiles/r04_ln_bench.txt)
#define VJ_LNB(NCHV, CSV)                                                                                                     \
  hipLaunchKernelGGL((layernorm_bwd_kernel<NCHV, CSV, true>), dim3((unsigned)nb), dim3(256), 0, stream, (const bf16_t*)dy_bf16, \
                     (const bf16_t*)x_bf16, gamma, mean, rstd, (const bf16_t*)dres_bf16, (bf16_t*)dx_bf16, (float*)ws,         \
                     rows, (int)D)
  if (cs) {
    if (D <= 512) VJ_LNB(1, true);
    else if (D <= 1024) VJ_LNB(2, true);
    else if (D <= 1536) VJ_LNB(3, true);
    else VJ_LNB(4, true);
  } else {
    if (D <= 512) VJ_LNB(1, false);
    else if (D <= 1024) VJ_LNB(2, false);
    else if (D <= 1536) VJ_LNB(3, false);
    else VJ_LNB(4, false);
  }
#undef VJ_LNB
  VJ_LAUNCH_CHECK("vj_layernorm_bwd");
  *nb_out = nb;
  return 0;
}

// dgamma/dbeta (and dxsum, nullable: column sums of dx) : out = alpha * sum + beta_acc * out   (beta_acc = 1 accumulates)
extern "C" int vj_layernorm_bwd_colsum(const void* dy_bf16, const void* x_bf16, const float* gamma, const float* mean,
                                       const float* rstd, const void* dres_bf16, void* dx_bf16, float* dgamma,
                                       float* dbeta, float* dxsum, float alpha, float beta_acc, int64_t rows, int64_t D,
                                       void* ws, int64_t ws_bytes, hipStream_t stream) {
  int64_t nb = 0;
  if (int rc = vj_layernorm_bwd_partials(dy_bf16, x_bf16, gamma, mean, rstd, dres_bf16, dx_bf16, dxsum != nullptr, rows, D, ws,
                                         ws_bytes, &nb, stream))
    return rc;
  if (nb == 0) return 0;
  float* outs[3] = {dgamma, dbeta, dxsum};
  return vj_reduce_partials_multi((const float*)ws, outs, dxsum != nullptr ? 3 : 2, nb, D, alpha, beta_acc, stream);   // ONE launch
}

extern "C" int vj_layernorm_bwd(const void* dy_bf16, const void* x_bf16, const float* gamma, const float* mean,
                                const float* rstd, const void* dres_bf16, void* dx_bf16, float* dgamma, float* dbeta,
                                float alpha, float beta_acc, int64_t rows, int64_t D, void* ws, int64_t ws_bytes,
                                hipStream_t stream) {
  return vj_layernorm_bwd_colsum(dy_bf16, x_bf16, gamma, mean, rstd, dres_bf16, dx_bf16, dgamma, dbeta, nullptr, alpha,
                                 beta_acc, rows, D, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// target_rows: h[b,k,:] = LN_noaffine_{eps2}( LN_{gamma,beta,eps1}( x[b, idx[b,k], :] ) )   fp32 out
// (final encoder norm + F.layer_norm + apply_masks fused; only the K predicted rows are ever normalised)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void target_rows_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const int64_t* __restrict__ idx, float* __restrict__ h,
                                                          int64_t B, int64_t N, int64_t K, int D, float eps1,
                                                          float eps2) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const float invD = 1.0f / (float)D;
  for (int64_t r = wave; r < B * K; r += nw) {
    const int64_t b = r / K;
    const bf16_t* xp = x + (b * N + idx[r]) * D;
    float v[LN_MAX_CHUNKS][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
        load8(xp + c, v[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) s += v[i][j];
      }
    }
    float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float d = v[i][j] - mean;
          q += d * d;
        }
      }
    }
    float rstd = rsqrtf(wave_sum(q) * invD + eps1);
    s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          v[i][j] = (v[i][j] - mean) * rstd * gamma[c + j] + beta[c + j];
          s += v[i][j];
        }
      }
    }
    mean = wave_sum(s) * invD;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float d = v[i][j] - mean;
          q += d * d;
        }
      }
    }
    rstd = rsqrtf(wave_sum(q) * invD + eps2);
    float* hp = h + r * D;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; i++) {
      const int c = lane * 8 + i * 512;
      if (c < D) {
        float4 o0, o1;
        o0.x = (v[i][0] - mean) * rstd;
        o0.y = (v[i][1] - mean) * rstd;
        o0.z = (v[i][2] - mean) * rstd;
        o0.w = (v[i][3] - mean) * rstd;
        o1.x = (v[i][4] - mean) * rstd;
        o1.y = (v[i][5] - mean) * rstd;
        o1.z = (v[i][6] - mean) * rstd;
        o1.w = (v[i][7] - mean) * rstd;
        *(float4*)(hp + c) = o0;
        *(float4*)(hp + c + 4) = o1;
      }
    }
  }
}

extern "C" int vj_target_rows(const void* x_bf16, const float* gamma, const float* beta, const int64_t* idx,
                              float* h, int64_t B, int64_t N, int64_t K, int64_t D, float eps_norm, float eps_ln,
                              hipStream_t stream) {
  VJ_CHECK_ARG(D % 8 == 0 && D <= 512 * LN_MAX_CHUNKS, "vj_target_rows: D=%ld unsupported", (long)D);
  if (B * K == 0) return 0;
  hipLaunchKernelGGL(target_rows_kernel, dim3(ln_grid(B * K)), dim3(256), 0, stream, (const bf16_t*)x_bf16, gamma,
                     beta, idx, h, B, N, K, (int)D, eps_norm, eps_ln);
  VJ_LAUNCH_CHECK("vj_target_rows");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// latent_loss: sum |z-h|^p / p over all elements (z bf16, h fp32), deterministic two-stage reduction,
// optionally writing dz = sign(z-h)*|z-h|^(p-1) * gscale (bf16) in the same pass.
// part[blk] holds the block sums; finish kernel folds them:  out[slot] = scale * sum.
// ---------------------------------------------------------------------------------------------
#define LOSS_BLOCKS 512
__global__ __launch_bounds__(256) void latent_loss_kernel(const bf16_t* __restrict__ z, const float* __restrict__ h,
                                                          bf16_t* __restrict__ dz, float* __restrict__ part,
                                                          int64_t n8, float p, float gscale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n8; q += (int64_t)gridDim.x * 256) {
    float zv[8];
    load8(z + q * 8, zv);
    const float4 h0 = *(const float4*)(h + q * 8);
    const float4 h1 = *(const float4*)(h + q * 8 + 4);
    const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float d = zv[j] - hv[j];
      const float a = fabsf(d);
      if (p == 1.0f) {
        acc += a;
        g[j] = (d > 0.f) ? gscale : ((d < 0.f) ? -gscale : 0.f);
      } else {
        acc += __powf(a, p) / p;
        const float m = (a > 0.f) ? __powf(a, p - 1.0f) : 0.f;
        g[j] = (d > 0.f) ? m * gscale : -m * gscale;
      }
    }
    if (dz) store8(dz + q * 8, g);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void scalar_finish_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ out,
                                     int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (red[0] + red[1] + red[2] + red[3]) * scale;
    *out = accumulate ? (*out + v) : v;
  }
}

extern "C" int64_t vj_latent_loss_ws_bytes(void) { return LOSS_BLOCKS * 4; }

// loss_out (device scalar) = [accumulate ? loss_out : 0] + out_scale * sum(|z-h|^p / p)
extern "C" int vj_latent_loss(const void* z_bf16, const float* h, void* dz_bf16, int64_t numel, float p,
                              float gscale, float out_scale, int accumulate, float* loss_out, void* ws,
                              int64_t ws_bytes, hipStream_t stream) {
  VJ_CHECK_ARG(numel % 8 == 0, "vj_latent_loss: numel=%ld must be a multiple of 8", (long)numel);
  VJ_CHECK_ARG(ws_bytes >= vj_latent_loss_ws_bytes(), "vj_latent_loss: workspace too small");
  VJ_CHECK_ARG(p > 0.f, "vj_latent_loss: loss_exp must be > 0");
  if (numel == 0) return 0;
  hipLaunchKernelGGL(latent_loss_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, stream, (const bf16_t*)z_bf16, h,
                     (bf16_t*)dz_bf16, (float*)ws, numel / 8, p, gscale);
  VJ_LAUNCH_CHECK("vj_latent_loss");
  hipLaunchKernelGGL(scalar_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, LOSS_BLOCKS, out_scale,
                     loss_out, accumulate);
  VJ_LAUNCH_CHECK("vj_latent_loss(finish)");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// token_pstd: pstd[b,d] (+)= sqrt(unbiased_var_k z[b,k,d] + 1e-4)   (reg_fn, train.py:448-449)
// one block per (b, 256-column slab); two-pass over the K rows for accuracy.
// ---------------------------------------------------------------------------------------------
// one workgroup per (b, 64-column slab): 4 waves split the K rows, each lane owns one column pair... 8 columns per
// thread (16-byte loads), shifted single-pass sums (shift = first row) combined across the 8 row-lanes in LDS.
__global__ __launch_bounds__(256) void token_pstd_kernel(const bf16_t* __restrict__ z, float* __restrict__ pstd,
                                                         float* __restrict__ stats, int64_t K, int D,
                                                         int accumulate) {
  __shared__ float red[2][32][65];
  const int64_t b = blockIdx.y;
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;       // 8 column groups of 8, 32 row lanes
  const int d0 = blockIdx.x * 64 + cg * 8;
  const bf16_t* zp = z + b * K * D;
  float s[8], q[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; j++) s[j] = q[j] = sh[j] = 0.f;
  if (d0 < D) {
    load8(zp + d0, sh);                                         // shift by row 0: well-conditioned single pass
    for (int64_t k = rl; k < K; k += 32) {
      float v[8];
      load8(zp + k * D + d0, v);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float t = v[j] - sh[j];
        s[j] += t;
        q[j] += t * t;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    red[0][rl][cg * 8 + j] = s[j];
    red[1][rl][cg * 8 + j] = q[j];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = threadIdx.x, d = blockIdx.x * 64 + c;
    if (d < D) {
      float ss = 0.f, qq = 0.f;
      for (int r = 0; r < 32; r++) {
        ss += red[0][r][c];
        qq += red[1][r][c];
      }
      const float var = (qq - ss * ss / (float)K) / (float)(K - 1);   // unbiased, shift-invariant
      const float v = sqrtf(fmaxf(var, 0.f) + 1e-4f);
      float* o = pstd + b * D + d;
      *o = accumulate ? (*o + v) : v;
      if (stats) {   // per-(b,d) token mean and sqrt(var + eps) of THIS mask, for reg_grad
        stats[(b * D + d) * 2] = bf2f(zp[d]) + ss / (float)K;
        stats[(b * D + d) * 2 + 1] = v;
      }
    }
  }
}

// reg = mean(relu(1 - pstd_sum / n_masks))
__global__ __launch_bounds__(256) void reg_finish_kernel(const float* __restrict__ pstd, int64_t n, float inv_masks,
                                                         float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += fmaxf(0.f, 1.0f - pstd[i] * inv_masks);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

extern "C" int vj_token_pstd(const void* z_bf16, float* pstd, float* stats, int64_t B, int64_t K, int64_t D,
                             int accumulate, hipStream_t stream) {
  VJ_CHECK_ARG(K >= 2, "vj_token_pstd: need at least 2 tokens for an unbiased variance (K=%ld)", (long)K);
  if (B * D == 0) return 0;
  VJ_CHECK_ARG(D % 8 == 0, "vj_token_pstd: D must be a multiple of 8");
  hipLaunchKernelGGL(token_pstd_kernel, dim3((unsigned)cdiv64(D, 64), (unsigned)B), dim3(256), 0, stream,
                     (const bf16_t*)z_bf16, pstd, stats, K, (int)D, accumulate);
  VJ_LAUNCH_CHECK("vj_token_pstd");
  return 0;
}

extern "C" int vj_reg_finish(const float* pstd_sum, int64_t n, int64_t n_masks, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(reg_finish_kernel, dim3(1), dim3(256), 0, stream, pstd_sum, n, 1.0f / (float)n_masks, out);
  VJ_LAUNCH_CHECK("vj_reg_finish");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// reg_grad: dz[b,k,d] += coef * d/dz mean_{b,d} relu(1 - pstd_avg[b,d]),  pstd_avg = pstd_sum / n_masks
//   = -coef / (B*D*n_masks) * 1[pstd_avg < 1] * (z - mean) / ((K-1) * sqrt(var + eps))       (train.py:448-459)
// dz holds the latent-loss gradient in units of 1/gscale (see vj_latent_loss); `coef` is pre-divided accordingly.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reg_grad_kernel(const bf16_t* __restrict__ z, const float* __restrict__ pstd_sum,
                                                       const float* __restrict__ stats, bf16_t* __restrict__ dz,
                                                       int64_t B, int64_t K, int D, float inv_masks, float coef) {
  const int64_t n8 = B * K * D / 8;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n8; q += (int64_t)gridDim.x * 256) {
    const int64_t e = q * 8;
    const int d0 = (int)(e % D);
    const int64_t b = e / ((int64_t)K * D);
    float zv[8], gv[8];
    load8(z + e, zv);
    load8(dz + e, gv);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int64_t bd = b * D + d0 + j;
      const float active = (pstd_sum[bd] * inv_masks < 1.0f) ? 1.0f : 0.f;
      gv[j] -= coef * active * (zv[j] - stats[bd * 2]) / ((float)(K - 1) * stats[bd * 2 + 1]);
    }
    store8(dz + e, gv);
  }
}

extern "C" int vj_reg_grad(const void* z_bf16, const float* pstd_sum, const float* stats, void* dz_bf16, int64_t B,
                           int64_t K, int64_t D, int64_t n_masks, float coef, hipStream_t stream) {
  VJ_CHECK_ARG(D % 8 == 0 && K >= 2, "vj_reg_grad: need D %% 8 == 0 and K >= 2");
  if (B * K * D == 0) return 0;
  int64_t g = cdiv64(B * K * D / 8, 256);
  if (g > 256 * 8) g = 256 * 8;
  hipLaunchKernelGGL(reg_grad_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)z_bf16, pstd_sum, stats,
                     (bf16_t*)dz_bf16, B, K, (int)D, 1.0f / (float)n_masks, coef);
  VJ_LAUNCH_CHECK("vj_reg_grad");
  return 0;
}
