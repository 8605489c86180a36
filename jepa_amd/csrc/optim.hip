// Fused parameter update of the V-JEPA step over flat fp32 arenas:
//   AdamW (decoupled weight decay, bias correction)  -> torch.optim.AdamW semantics, app/vjepa/utils.py:193
//   EMA of the target encoder  theta_k = m*theta_k + (1-m)*theta_q        app/vjepa/train.py:483-487
//   bf16 re-cast of both weight sets (the operands of next step's MFMA GEMMs)
// One pass, 28 B/param of mandatory traffic + 2-4 B/param of bf16 shadows. HBM-bound.
#include "common.hpp"
#include "options.hpp"

struct AdamArgs {
  float* p;         // master weights (fp32)
  const float* g;   // gradients (fp32)
  float* m;         // exp_avg
  float* v;         // exp_avg_sq
  bf16_t* p_bf16;   // bf16 shadow of p (nullable)
  float* tgt;       // EMA target weights (nullable)
  bf16_t* tgt_bf16; // bf16 shadow of tgt (nullable)
  int64_t n;
  float lr, wd, beta1, beta2, eps;
  float bc1, bc2_sqrt;  // 1-beta1^t, sqrt(1-beta2^t)
  float gscale;         // gradient pre-scale (clip coefficient / all-reduce mean)
  float ema;            // momentum m
  // device-side guard (vj_adamw_ema_guarded; all nullable / 0 for the plain entry point):
  const float* gstat;   // [sumsq_0, nonfinite_0, sumsq_1, nonfinite_1] of the (summed) gradients, from vj_sqnorm_f32
  int sel;              // which sumsq belongs to this parameter range (clip_grad_norm_ is per module, train.py:469-470)
  float clip;           // max norm (<= 0: no clipping)
  float norm_scale;     // gradient norm = sqrt(sumsq) * norm_scale (1/world for summed data-parallel gradients)
  const float* step_dev;  // device step counter t (already advanced by vj_step_advance); bias corrections from it
};

__global__ __launch_bounds__(256) void adamw_ema_kernel(AdamArgs a) {
  const int64_t n4 = a.n >> 2;
  bool skip = false;
  if (a.gstat) {
    // GradScaler semantics (scaler.step, train.py:471): any non-finite gradient anywhere -> no optimizer step at all;
    // the EMA of train.py:483-487 still runs (against the unchanged weights).  Decided on the device: no host sync.
    skip = (a.gstat[1] + a.gstat[3]) > 0.f;
    if (a.clip > 0.f) {   // torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (total_norm + 1e-6))
      const float norm = sqrtf(a.gstat[2 * a.sel]) * a.norm_scale;
      a.gscale *= fminf(1.0f, a.clip / (norm + 1e-6f));
    }
  }
  if (a.step_dev) {
    const double t = (double)*a.step_dev;
    a.bc1 = (float)(1.0 - pow((double)a.beta1, t));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, t));
  }
  if (skip) {
    if (!a.tgt) return;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
      const float4 p = ((const float4*)a.p)[q];
      float4 t = ((float4*)a.tgt)[q];
      t.x = t.x * a.ema + (1.0f - a.ema) * p.x;
      t.y = t.y * a.ema + (1.0f - a.ema) * p.y;
      t.z = t.z * a.ema + (1.0f - a.ema) * p.z;
      t.w = t.w * a.ema + (1.0f - a.ema) * p.w;
      ((float4*)a.tgt)[q] = t;
      if (a.tgt_bf16) {
        u32x2_t w;
        w[0] = pack_bf2(t.x, t.y);
        w[1] = pack_bf2(t.z, t.w);
        ((u32x2_t*)a.tgt_bf16)[q] = w;
      }
    }
    return;
  }
  const float step = a.lr / a.bc1;
  const float decay = 1.0f - a.lr * a.wd;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    float4 p = ((float4*)a.p)[q];
    const float4 g4 = ((const float4*)a.g)[q];
    float4 m = ((float4*)a.m)[q];
    float4 v = ((float4*)a.v)[q];
    float pp[4] = {p.x, p.y, p.z, p.w};
    const float gg[4] = {g4.x * a.gscale, g4.y * a.gscale, g4.z * a.gscale, g4.w * a.gscale};
    float mm[4] = {m.x, m.y, m.z, m.w};
    float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      pp[i] *= decay;                                        // p.mul_(1 - lr*wd)
      mm[i] = mm[i] + (gg[i] - mm[i]) * (1.0f - a.beta1);    // exp_avg.lerp_(g, 1-beta1)
      vv[i] = vv[i] * a.beta2 + (1.0f - a.beta2) * gg[i] * gg[i];
      const float denom = sqrtf(vv[i]) / a.bc2_sqrt + a.eps;
      pp[i] -= step * (mm[i] / denom);
    }
    ((float4*)a.p)[q] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    ((float4*)a.m)[q] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    ((float4*)a.v)[q] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (a.p_bf16) {
      u32x2_t w;
      w[0] = pack_bf2(pp[0], pp[1]);
      w[1] = pack_bf2(pp[2], pp[3]);
      ((u32x2_t*)a.p_bf16)[q] = w;
    }
    if (a.tgt) {
      const float4 t4 = ((float4*)a.tgt)[q];
      float tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
      for (int i = 0; i < 4; i++) tt[i] = tt[i] * a.ema + (1.0f - a.ema) * pp[i];
      ((float4*)a.tgt)[q] = make_float4(tt[0], tt[1], tt[2], tt[3]);
      if (a.tgt_bf16) {
        u32x2_t w;
        w[0] = pack_bf2(tt[0], tt[1]);
        w[1] = pack_bf2(tt[2], tt[3]);
        ((u32x2_t*)a.tgt_bf16)[q] = w;
      }
    }
  }
}

static inline int flat_grid(int64_t n_items) {
  int64_t g = cdiv64(n_items, 256);
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int vj_adamw_ema(float* p, const float* g, float* exp_avg, float* exp_avg_sq, void* p_bf16, float* tgt,
                            void* tgt_bf16, int64_t n, float lr, float wd, float beta1, float beta2, float eps,
                            int64_t step, float gscale, float ema, hipStream_t stream) {
  VJ_CHECK_ARG(n % 4 == 0, "vj_adamw_ema: segment length %ld must be a multiple of 4 (pad the arena)", (long)n);
  VJ_CHECK_ARG(step >= 1, "vj_adamw_ema: step must be >= 1");
  if (n == 0) return 0;
  AdamArgs a;
  a.p = p; a.g = g; a.m = exp_avg; a.v = exp_avg_sq; a.p_bf16 = (bf16_t*)p_bf16; a.tgt = tgt;
  a.tgt_bf16 = (bf16_t*)tgt_bf16; a.n = n; a.lr = lr; a.wd = wd; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.gscale = gscale; a.ema = ema;
  a.gstat = nullptr; a.sel = 0; a.clip = 0.f; a.norm_scale = 1.f; a.step_dev = nullptr;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, a);
  VJ_LAUNCH_CHECK("vj_adamw_ema");
  return 0;
}

// Device-guarded form: skip-on-non-finite, clip coefficient and the Adam step count are all read from device memory,
// so the host never synchronises between backward and update (the reference pays a float() in clip_grad_norm_ and
// an inf-check sync inside GradScaler.step, train.py:465-472).
__global__ void step_advance_kernel(const float* __restrict__ gstat, float* __restrict__ step_dev) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !((gstat[1] + gstat[3]) > 0.f)) *step_dev += 1.0f;
}
extern "C" int vj_step_advance(const float* gstat, float* step_dev, hipStream_t stream) {
  VJ_CHECK_ARG(gstat != nullptr && step_dev != nullptr, "vj_step_advance: null pointer");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, stream, gstat, step_dev);
  VJ_LAUNCH_CHECK("vj_step_advance");
  return 0;
}
extern "C" int vj_adamw_ema_guarded(float* p, const float* g, float* exp_avg, float* exp_avg_sq, void* p_bf16, float* tgt,
                                    void* tgt_bf16, int64_t n, float lr, float wd, float beta1, float beta2, float eps,
                                    float gscale, float ema, const float* gstat, int sel, float clip, float norm_scale,
                                    const float* step_dev, hipStream_t stream) {
  VJ_CHECK_ARG(n % 4 == 0, "vj_adamw_ema_guarded: segment length %ld must be a multiple of 4 (pad the arena)", (long)n);
  VJ_CHECK_ARG(gstat != nullptr && step_dev != nullptr && (sel == 0 || sel == 1), "vj_adamw_ema_guarded: bad guard arguments");
  if (n == 0) return 0;
  AdamArgs a;
  a.p = p; a.g = g; a.m = exp_avg; a.v = exp_avg_sq; a.p_bf16 = (bf16_t*)p_bf16; a.tgt = tgt;
  a.tgt_bf16 = (bf16_t*)tgt_bf16; a.n = n; a.lr = lr; a.wd = wd; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = 1.f; a.bc2_sqrt = 1.f; a.gscale = gscale; a.ema = ema;
  a.gstat = gstat; a.sel = sel; a.clip = clip; a.norm_scale = norm_scale; a.step_dev = step_dev;
  // (a cap on the workgroup count -- fewer CUs for the update while it runs beside the next step's forward -- was an option in round 5: level)
  const int grid = flat_grid(n / 4);
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(grid), dim3(256), 0, stream, a);
  VJ_LAUNCH_CHECK("vj_adamw_ema_guarded");
  return 0;
}

// EMA alone (frozen tensors such as pos_embed ride along in the reference loop, train.py:486)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ tgt, const float* __restrict__ src,
                                                  bf16_t* __restrict__ tgt_bf16, int64_t n4, float m) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    const float4 s = ((const float4*)src)[q];
    float4 t = ((float4*)tgt)[q];
    t.x = t.x * m + (1.0f - m) * s.x;
    t.y = t.y * m + (1.0f - m) * s.y;
    t.z = t.z * m + (1.0f - m) * s.z;
    t.w = t.w * m + (1.0f - m) * s.w;
    ((float4*)tgt)[q] = t;
    if (tgt_bf16) {
      u32x2_t w;
      w[0] = pack_bf2(t.x, t.y);
      w[1] = pack_bf2(t.z, t.w);
      ((u32x2_t*)tgt_bf16)[q] = w;
    }
  }
}

extern "C" int vj_ema_update(float* tgt, const float* src, void* tgt_bf16, int64_t n, float m, hipStream_t stream) {
  VJ_CHECK_ARG(n % 4 == 0, "vj_ema_update: n must be a multiple of 4");
  if (n == 0) return 0;
  hipLaunchKernelGGL(ema_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, tgt, src, (bf16_t*)tgt_bf16, n / 4, m);
  VJ_LAUNCH_CHECK("vj_ema_update");
  return 0;
}

// fp32 -> bf16 cast of a flat arena
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                        int64_t n4) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    const float4 s = ((const float4*)src)[q];
    u32x2_t w;
    w[0] = pack_bf2(s.x, s.y);
    w[1] = pack_bf2(s.z, s.w);
    ((u32x2_t*)dst)[q] = w;
  }
}

extern "C" int vj_cast_f32_to_bf16(const float* src, void* dst_bf16, int64_t n, hipStream_t stream) {
  VJ_CHECK_ARG(n % 4 == 0, "vj_cast_f32_to_bf16: n must be a multiple of 4");
  if (n == 0) return 0;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, stream, src, (bf16_t*)dst_bf16, n / 4);
  VJ_LAUNCH_CHECK("vj_cast_f32_to_bf16");
  return 0;
}

// sum of squares of a flat fp32 arena -> out[0] (+= if accumulate); also counts non-finite values in out[1]
#define SQ_BLOCKS 1024
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, int64_t n4, float* __restrict__ part) {
  __shared__ float red[2][4];
  float s = 0.f, bad = 0.f;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    const float4 v = ((const float4*)g)[q];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    bad += (float)(!isfinite(v.x)) + (float)(!isfinite(v.y)) + (float)(!isfinite(v.z)) + (float)(!isfinite(v.w));
  }
  s = wave_sum(s);
  bad = wave_sum(bad);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    part[SQ_BLOCKS + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}
__global__ __launch_bounds__(256) void sqnorm_finish_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            int accumulate) {
  __shared__ float red[2][4];
  float s = 0.f, bad = 0.f;
  for (int i = threadIdx.x; i < SQ_BLOCKS; i += 256) {
    s += part[i];
    bad += part[SQ_BLOCKS + i];
  }
  s = wave_sum(s);
  bad = wave_sum(bad);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    out[0] = accumulate ? out[0] + a : a;
    out[1] = accumulate ? out[1] + b : b;
  }
}
extern "C" int64_t vj_sqnorm_ws_bytes(void) { return 2 * SQ_BLOCKS * 4; }
extern "C" int vj_sqnorm_f32(const float* g, int64_t n, float* out2, int accumulate, void* ws, int64_t ws_bytes,
                             hipStream_t stream) {
  VJ_CHECK_ARG(n % 4 == 0, "vj_sqnorm_f32: n must be a multiple of 4");
  VJ_CHECK_ARG(ws_bytes >= vj_sqnorm_ws_bytes(), "vj_sqnorm_f32: workspace too small");
  hipLaunchKernelGGL(sqnorm_kernel, dim3(SQ_BLOCKS), dim3(256), 0, stream, g, n / 4, (float*)ws);
  VJ_LAUNCH_CHECK("vj_sqnorm_f32");
  hipLaunchKernelGGL(sqnorm_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, out2, accumulate);
  VJ_LAUNCH_CHECK("vj_sqnorm_f32(finish)");
  return 0;
}

// Per-tensor statistics of the gradient / Adam-moment arenas in ONE launch (reference: grad_logger + adamw_logger,
// src/utils/logging.py:91-118, which cost one float() host sync per tensor, ~1000 per step).  desc[t] = {offset, numel}
// (elements, offset % 4 == 0 -- arena slots are 64-aligned); out[t][c][0..2] = partial sums over chunk c of
// {g^2, |exp_avg|, |exp_avg_sq|}; the caller adds the GS_CHUNKS partials (a 30 KB copy, only when a log line is due).
#define GS_CHUNKS 8
__global__ __launch_bounds__(256) void grad_stats_multi_kernel(const float* __restrict__ G, const float* __restrict__ M1,
                                                               const float* __restrict__ M2,
                                                               const int64_t* __restrict__ desc, float* __restrict__ out) {
  __shared__ float red[3][4];
  const int t = blockIdx.x, c = blockIdx.y;
  const int64_t off = desc[2 * t], n = desc[2 * t + 1];
  const int64_t per = cdiv64(cdiv64(n, GS_CHUNKS), 4) * 4;   // chunk length, multiple of 4
  const int64_t lo = (int64_t)c * per, hi = lo + per < n ? lo + per : n;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const int64_t n4 = hi > lo ? (hi - lo) >> 2 : 0;
  for (int64_t q = threadIdx.x; q < n4; q += 256) {
    const int64_t e = off + lo + 4 * q;
    const float4 g = *(const float4*)(G + e);
    s0 += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
    if (M1) {
      const float4 a = *(const float4*)(M1 + e);
      const float4 b = *(const float4*)(M2 + e);
      s1 += fabsf(a.x) + fabsf(a.y) + fabsf(a.z) + fabsf(a.w);
      s2 += fabsf(b.x) + fabsf(b.y) + fabsf(b.z) + fabsf(b.w);
    }
  }
  for (int64_t e = lo + 4 * n4 + threadIdx.x; e < hi; e += 256) {   // tail (numel % 4 != 0 only in the last chunk)
    const float g = G[off + e];
    s0 += g * g;
    if (M1) {
      s1 += fabsf(M1[off + e]);
      s2 += fabsf(M2[off + e]);
    }
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s0;
    red[1][threadIdx.x >> 6] = s1;
    red[2][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    out[((int64_t)t * GS_CHUNKS + c) * 3 + threadIdx.x] =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
extern "C" int64_t vj_grad_stats_chunks(void) { return GS_CHUNKS; }
extern "C" int vj_grad_stats_multi(const float* G, const float* M1, const float* M2, const int64_t* desc, int64_t n_tensors,
                                   float* out, hipStream_t stream) {
  VJ_CHECK_ARG(G != nullptr && desc != nullptr && out != nullptr, "vj_grad_stats_multi: null pointer");
  VJ_CHECK_ARG((M1 == nullptr) == (M2 == nullptr), "vj_grad_stats_multi: pass both moment arenas or neither");
  VJ_CHECK_ARG(n_tensors >= 0 && n_tensors < 65536, "vj_grad_stats_multi: bad tensor count");
  if (n_tensors == 0) return 0;
  hipLaunchKernelGGL(grad_stats_multi_kernel, dim3((unsigned)n_tensors, GS_CHUNKS), dim3(256), 0, stream, G, M1, M2, desc,
                     out);
  VJ_LAUNCH_CHECK("vj_grad_stats_multi");
  return 0;
}
