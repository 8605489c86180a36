// Row-granular HBM-bound kernels of the V-JEPA step: bit-exact mask gather/scatter of token rows,
// tubelet packing of fp32 clips into bf16 patch rows, positional-embedding add, predictor token
// assembly, bf16 transposes (wgrad operands) and column sums (bias grads).
//
// Reference behaviour restated (never copied):
//   apply_masks                      src/masks/utils.py:11-23      (gather of kept token rows)
//   PatchEmbed3D token/K ordering    src/models/utils/patch_embed.py:31-57
//   x += pos_embed                   src/models/vision_transformer.py:172-174
//   predictor token assembly         src/models/predictor.py:194-221
#include "common.hpp"
#include "../../include/vjepa_hip.h"

// ---------------------------------------------------------------------------------------------
// gather_rows: dst[b,k,:] = src[b*src_bstride + idx[b,k], :]   (payload moved verbatim -> bit exact)
// one wave per row, 16 B per lane per trip (row_bytes % 16 == 0) or 4 B per lane (row_bytes % 4 == 0)
// ---------------------------------------------------------------------------------------------
template <typename VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                          const int64_t* __restrict__ idx, int64_t rows, int64_t K,
                                                          int64_t row_bytes, int64_t src_bstride_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int64_t nvec = row_bytes / (int64_t)sizeof(VEC);
  for (int64_t r = wave; r < rows; r += nw) {
    const int64_t b = r / K;
    const int64_t s = b * src_bstride_rows + idx[r];
    const VEC* sp = (const VEC*)(src + s * row_bytes);
    VEC* dp = (VEC*)(dst + r * row_bytes);
    for (int64_t v = lane; v < nvec; v += 64) dp[v] = sp[v];
  }
}

// scatter_rows: dst[b, idx[b,k], :] = src[b,k,:]  (dst pre-zeroed; indices unique per b as produced by
// the collator, multiblock3d.py:185-186) -- backward of gather_rows.
template <typename VEC>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                           const int64_t* __restrict__ idx, int64_t rows, int64_t K,
                                                           int64_t row_bytes, int64_t N) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int64_t nvec = row_bytes / (int64_t)sizeof(VEC);
  for (int64_t r = wave; r < rows; r += nw) {
    const int64_t b = r / K;
    const int64_t d = b * N + idx[r];
    const VEC* sp = (const VEC*)(src + r * row_bytes);
    VEC* dp = (VEC*)(dst + d * row_bytes);
    for (int64_t v = lane; v < nvec; v += 64) dp[v] = sp[v];
  }
}

static inline int rows_grid(int64_t rows) {
  int64_t g = cdiv64(rows, 4);
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int vj_gather_rows(const void* src, void* dst, const int64_t* idx, int64_t B, int64_t K, int64_t row_bytes,
                              int64_t src_batch_stride_rows, hipStream_t stream) {
  VJ_CHECK_ARG(B >= 0 && K >= 0 && row_bytes > 0, "vj_gather_rows: bad dims B=%ld K=%ld row_bytes=%ld", (long)B,
               (long)K, (long)row_bytes);
  VJ_CHECK_ARG(row_bytes % 4 == 0, "vj_gather_rows: row_bytes=%ld must be a multiple of 4", (long)row_bytes);
  const int64_t rows = B * K;
  if (rows == 0) return 0;
  const bool v16 = (row_bytes % 16 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
  if (v16)
    hipLaunchKernelGGL(gather_rows_kernel<u32x4_t>, dim3(rows_grid(rows)), dim3(256), 0, stream, (const char*)src,
                       (char*)dst, idx, rows, K, row_bytes, src_batch_stride_rows);
  else
    hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(rows_grid(rows)), dim3(256), 0, stream, (const char*)src,
                       (char*)dst, idx, rows, K, row_bytes, src_batch_stride_rows);
  VJ_LAUNCH_CHECK("vj_gather_rows");
  return 0;
}

extern "C" int vj_scatter_rows(const void* src, void* dst, const int64_t* idx, int64_t B, int64_t N, int64_t K,
                               int64_t row_bytes, hipStream_t stream) {
  VJ_CHECK_ARG(B >= 0 && K >= 0 && N >= 0 && row_bytes > 0, "vj_scatter_rows: bad dims");
  VJ_CHECK_ARG(row_bytes % 4 == 0, "vj_scatter_rows: row_bytes=%ld must be a multiple of 4", (long)row_bytes);
  if (B * N == 0) return 0;
  hipError_t e = hipMemsetAsync(dst, 0, (size_t)(B * N * row_bytes), stream);
  if (e != hipSuccess) {
    vj_set_error("vj_scatter_rows: memset failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  const int64_t rows = B * K;
  if (rows == 0) return 0;
  const bool v16 = (row_bytes % 16 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
  if (v16)
    hipLaunchKernelGGL(scatter_rows_kernel<u32x4_t>, dim3(rows_grid(rows)), dim3(256), 0, stream, (const char*)src,
                       (char*)dst, idx, rows, K, row_bytes, N);
  else
    hipLaunchKernelGGL(scatter_rows_kernel<uint32_t>, dim3(rows_grid(rows)), dim3(256), 0, stream, (const char*)src,
                       (char*)dst, idx, rows, K, row_bytes, N);
  VJ_LAUNCH_CHECK("vj_scatter_rows");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// tubelet_pack: fp32 clips [B,C,T,H,W] -> bf16 patch rows [B,K,C*tub*p*p]; row k of clip b is token
// n = idx ? idx[b,k] : k, n -> (t',h',w') row-major (flatten(2).transpose(1,2), patch_embed.py:56),
// element order (c,dt,dh,dw) = Conv3d weight order [D,C,tub,p,p].  8 pixels per thread.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tubelet_pack_kernel(const float* __restrict__ clips, bf16_t* __restrict__ out,
                                                           const int64_t* __restrict__ idx, int64_t B, int C, int T,
                                                           int H, int W, int tub, int p, int64_t K) {
  const int gh = H / p, gw = W / p;
  const int kdim = C * tub * p * p;
  const int cpr = kdim / 8;  // 16-byte output chunks per row
  const int64_t total = B * K * cpr;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
    const int64_t row = q / cpr;
    const int e = (int)(q - row * cpr) * 8;
    const int64_t b = row / K;
    const int64_t n = idx ? idx[row] : (row - b * K);
    const int wq = (int)(n % gw), hq = (int)((n / gw) % gh), tq = (int)(n / ((int64_t)gw * gh));
    const int dw = e % p, dh = (e / p) % p, dt = (e / (p * p)) % tub, c = e / (p * p * tub);
    const float* s = clips + ((((b * C + c) * T + (tq * tub + dt)) * H + (hq * p + dh)) * (int64_t)W + wq * p + dw);
    const float4 lo = *(const float4*)s;
    const float4 hi = *(const float4*)(s + 4);
    u32x4_t o;
    o[0] = pack_bf2(lo.x, lo.y);
    o[1] = pack_bf2(lo.z, lo.w);
    o[2] = pack_bf2(hi.x, hi.y);
    o[3] = pack_bf2(hi.z, hi.w);
    *(u32x4_t*)(out + row * kdim + e) = o;
  }
}

extern "C" int vj_tubelet_pack(const float* clips, void* out_bf16, const int64_t* idx, int64_t B, int64_t C,
                               int64_t T, int64_t H, int64_t W, int64_t tubelet, int64_t patch, int64_t K,
                               hipStream_t stream) {
  VJ_CHECK_ARG(patch % 8 == 0 && W % 4 == 0, "vj_tubelet_pack: patch (%ld) must be a multiple of 8 and W%%4==0",
               (long)patch);
  VJ_CHECK_ARG(T % tubelet == 0 && H % patch == 0 && W % patch == 0, "vj_tubelet_pack: clip not divisible into tubelets");
  if (B * K == 0) return 0;
  const int64_t total = B * K * (C * tubelet * patch * patch / 8);
  int64_t g = cdiv64(total, 256);
  if (g > 256 * 32) g = 256 * 32;
  hipLaunchKernelGGL(tubelet_pack_kernel, dim3((int)g), dim3(256), 0, stream, clips, (bf16_t*)out_bf16, idx, B,
                     (int)C, (int)T, (int)H, (int)W, (int)tubelet, (int)patch, K);
  VJ_LAUNCH_CHECK("vj_tubelet_pack");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// add_pos: x[b,k,:] (bf16) += pos[idx ? idx[b,k] : k, :] (fp32), fp32 add, one rounding.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_pos_kernel(bf16_t* __restrict__ x, const float* __restrict__ pos,
                                                      const int64_t* __restrict__ idx, int64_t rows, int64_t K,
                                                      int D) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < rows; r += nw) {
    const int64_t n = idx ? idx[r] : (r % K);
    bf16_t* xp = x + r * D;
    const float* pp = pos + n * D;
    for (int c = lane * 8; c < D; c += 512) {
      u32x4_t v = *(u32x4_t*)(xp + c);
      const float4 p0 = *(const float4*)(pp + c);
      const float4 p1 = *(const float4*)(pp + c + 4);
      v[0] = pack_bf2(bf_lo(v[0]) + p0.x, bf_hi(v[0]) + p0.y);
      v[1] = pack_bf2(bf_lo(v[1]) + p0.z, bf_hi(v[1]) + p0.w);
      v[2] = pack_bf2(bf_lo(v[2]) + p1.x, bf_hi(v[2]) + p1.y);
      v[3] = pack_bf2(bf_lo(v[3]) + p1.z, bf_hi(v[3]) + p1.w);
      *(u32x4_t*)(xp + c) = v;
    }
  }
}

extern "C" int vj_add_pos(void* x_bf16, const float* pos, const int64_t* idx, int64_t B, int64_t K, int64_t D,
                          hipStream_t stream) {
  VJ_CHECK_ARG(D % 8 == 0, "vj_add_pos: D=%ld must be a multiple of 8", (long)D);
  const int64_t rows = B * K;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(add_pos_kernel, dim3(rows_grid(rows)), dim3(256), 0, stream, (bf16_t*)x_bf16, pos, idx, rows, K,
                     (int)D);
  VJ_LAUNCH_CHECK("vj_add_pos");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// pred_assemble_fwd: out[b, j, :] = j < Ke ? e[b,j,:] + pos[idx_e[b,j]] : tok[:] + pos[idx_p[b,j-Ke]]
//   e  = predictor_embed(z)   bf16 [B,Ke,Dp]     (predictor.py:194-200)
//   tok = mask_tokens[i]      fp32 [Dp]          (predictor.py:207-217)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pred_assemble_kernel(const bf16_t* __restrict__ e, const float* __restrict__ tok,
                                                            const float* __restrict__ pos,
                                                            const int64_t* __restrict__ idx_e,
                                                            const int64_t* __restrict__ idx_p,
                                                            bf16_t* __restrict__ out, int64_t B, int64_t Ke,
                                                            int64_t Kp, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int64_t S = Ke + Kp;
  for (int64_t r = wave; r < B * S; r += nw) {
    const int64_t b = r / S, j = r - b * S;
    const bool ctx = j < Ke;
    const int64_t n = ctx ? idx_e[b * Ke + j] : idx_p[b * Kp + (j - Ke)];
    const float* pp = pos + n * D;
    const bf16_t* ep = e + (b * Ke + j) * D;
    bf16_t* op = out + r * D;
    for (int c = lane * 8; c < D; c += 512) {
      float v[8];
      if (ctx) {
        const u32x4_t w = *(const u32x4_t*)(ep + c);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          v[2 * i] = bf_lo(w[i]);
          v[2 * i + 1] = bf_hi(w[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = tok[c + i];
      }
      u32x4_t o;
#pragma unroll
      for (int i = 0; i < 4; i++) o[i] = pack_bf2(v[2 * i] + pp[c + 2 * i], v[2 * i + 1] + pp[c + 2 * i + 1]);
      *(u32x4_t*)(op + c) = o;
    }
  }
}

extern "C" int vj_pred_assemble_fwd(const void* e_bf16, const float* mask_token, const float* pos,
                                    const int64_t* idx_e, const int64_t* idx_p, void* out_bf16, int64_t B,
                                    int64_t Ke, int64_t Kp, int64_t D, hipStream_t stream) {
  VJ_CHECK_ARG(D % 8 == 0, "vj_pred_assemble_fwd: D=%ld must be a multiple of 8", (long)D);
  const int64_t rows = B * (Ke + Kp);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(pred_assemble_kernel, dim3(rows_grid(rows)), dim3(256), 0, stream, (const bf16_t*)e_bf16,
                     mask_token, pos, idx_e, idx_p, (bf16_t*)out_bf16, B, Ke, Kp, (int)D);
  VJ_LAUNCH_CHECK("vj_pred_assemble_fwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// transpose_bf16: in [M,N] (ld_in) -> out [N, Mpad], out[n, m>=M] = 0.  64x64 tiles through LDS.
// Feeds the K-contiguous ("NT") MFMA GEMM with the wgrad operands dY^T and X^T.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                             int64_t M, int64_t N, int64_t ld_in, int64_t Mpad,
                                                             float* __restrict__ part) {
  __shared__ bf16_t tile[64][66];
  const int64_t m0 = (int64_t)blockIdx.x * 64, n0 = (int64_t)blockIdx.y * 64;
  const int t = threadIdx.x;
  // load: 64 rows x 8 chunks of 8 bf16
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int q = t + it * 256;
    const int r = q >> 3, c = (q & 7) * 8;
    u32x4_t v = {0, 0, 0, 0};
    if (m0 + r < M && n0 + c < N) v = *(const u32x4_t*)(in + (m0 + r) * ld_in + n0 + c);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      *(uint32_t*)&tile[r][c + 2 * i] = v[i];
    }
  }
  __syncthreads();
  if (part != nullptr && t < 64 && n0 + t < N) {  // fused bias-gradient partial: column sums of this 64-row tile
    float sum = 0.f;
#pragma unroll 16
    for (int r = 0; r < 64; r++) sum += bf2f(tile[r][t]);
    part[(int64_t)blockIdx.x * N + n0 + t] = sum;
  }
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int q = t + it * 256;
    const int n = q >> 3, mc = (q & 7) * 8;
    if (n0 + n < N && m0 + mc < Mpad) {
      u32x4_t o;
#pragma unroll
      for (int i = 0; i < 4; i++)
        o[i] = (uint32_t)tile[mc + 2 * i][n] | ((uint32_t)tile[mc + 2 * i + 1][n] << 16);
      *(u32x4_t*)(out + (n0 + n) * Mpad + m0 + mc) = o;
    }
  }
}

extern "C" int vj_transpose_bf16(const void* in, void* out, int64_t M, int64_t N, int64_t ld_in, int64_t Mpad,
                                 hipStream_t stream) {
  VJ_CHECK_ARG(N % 8 == 0 && ld_in % 8 == 0 && Mpad % 8 == 0 && Mpad >= M,
               "vj_transpose_bf16: need N,ld_in,Mpad multiples of 8 and Mpad>=M (M=%ld N=%ld ld=%ld Mpad=%ld)",
               (long)M, (long)N, (long)ld_in, (long)Mpad);
  if (N == 0 || Mpad == 0) return 0;
  dim3 grid((unsigned)cdiv64(Mpad, 64), (unsigned)cdiv64(N, 64));
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)in, (bf16_t*)out, M, N, ld_in,
                     Mpad, (float*)nullptr);
  VJ_LAUNCH_CHECK("vj_transpose_bf16");
  return 0;
}

extern "C" int vj_reduce_partials(const float* part, float* out, int64_t P, int64_t N, float alpha, float beta,
                                  hipStream_t stream);

// transpose + bias gradient in one pass over dY: out = in^T (zero padded), colsum[n] = alpha*sum_m in[m][n] + beta*colsum[n]
extern "C" int64_t vj_transpose_colsum_ws_bytes(int64_t M, int64_t N) { return cdiv64(((M + 63) / 64) * 64, 64) * N * 4; }

extern "C" int vj_transpose_colsum_bf16(const void* in, void* out, int64_t M, int64_t N, int64_t ld_in, int64_t Mpad,
                                        float* colsum, float alpha, float beta, void* ws, int64_t ws_bytes,
                                        hipStream_t stream) {
  VJ_CHECK_ARG(N % 8 == 0 && ld_in % 8 == 0 && Mpad % 8 == 0 && Mpad >= M, "vj_transpose_colsum_bf16: bad dims");
  const int64_t mt = cdiv64(Mpad, 64);
  VJ_CHECK_ARG(ws_bytes >= mt * N * 4, "vj_transpose_colsum_bf16: workspace too small");
  if (N == 0 || Mpad == 0) return 0;
  dim3 grid((unsigned)mt, (unsigned)cdiv64(N, 64));
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)in, (bf16_t*)out, M, N, ld_in,
                     Mpad, (float*)ws);
  VJ_LAUNCH_CHECK("vj_transpose_colsum_bf16");
  return vj_reduce_partials((const float*)ws, colsum, mt, N, alpha, beta, stream);
}

// ---------------------------------------------------------------------------------------------
// colsum_bf16: partial[p][n] = sum over the p-th row chunk of in[m][n] (rows m in [row_lo,row_hi) of
// each group of `group` rows -- used both for plain bias grads (group = M) and for the mask-token grad,
// which sums only the target rows j >= Ke of every [Ke+Kp]-row sample).  Deterministic two-stage sum.
// ---------------------------------------------------------------------------------------------
#define VJ_COLSUM_PARTS 256   // maximum number of row chunks (workspace sizing); the launcher picks 64 .. 256
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ in, float* __restrict__ part,
                                                          int64_t M, int64_t N, int64_t ld, int64_t group,
                                                          int64_t row_lo, int64_t row_hi, int parts) {
  __shared__ float red[8][256];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;  // 32 column groups of 8, 8 row lanes
  const int64_t n = (int64_t)blockIdx.x * 256 + cg * 8;
  const int64_t p = blockIdx.y;
  const int64_t rows_per = cdiv64(M, parts);
  const int64_t mbeg = p * rows_per, mend = (mbeg + rows_per < M) ? mbeg + rows_per : M;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n < N) {
    for (int64_t m = mbeg + rl; m < mend; m += 8) {
      const int64_t j = m % group;
      if (j < row_lo || j >= row_hi) continue;
      const u32x4_t v = *(const u32x4_t*)(in + m * ld + n);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        acc[2 * i] += bf_lo(v[i]);
        acc[2 * i + 1] += bf_hi(v[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) red[rl][cg * 8 + i] = acc[i];
  __syncthreads();
  const int c = threadIdx.x;
  const int64_t nn = (int64_t)blockIdx.x * 256 + c;
  if (nn < N) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; r++) s += red[r][c];
    part[p * N + nn] = s;
  }
}

// out[n] = alpha * sum_p part[p*stride + n] + (beta != 0 ? beta * out[n] : 0)
// one workgroup per 64 columns, 8 partial-lanes of 64 threads each (coalesced 256-B row reads, 8 in flight),
// then a fixed-order LDS combine -> deterministic.  The N columns may be split into up to three segments of `seg`
// columns with their own outputs (several reductions over one partial matrix in ONE launch: LayerNorm's
// dgamma | dbeta | column sum of dx); seg must be a multiple of 64.
struct ReduceOuts {
  float* o[3];
};
__global__ __launch_bounds__(512) void reduce_partials_kernel(const float* __restrict__ part, ReduceOuts outs, int64_t seg,
                                                              int64_t P, int64_t N, int64_t stride, float alpha,
                                                              float beta) {
  __shared__ float red[8][64];
  const int c = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 64 + c;
  float s = 0.f;
  if (n < N) {
#pragma unroll 8
    for (int64_t p = pl; p < P; p += 8) s += part[p * stride + n];
  }
  red[pl][c] = s;
  __syncthreads();
  if (pl == 0 && n < N) {
    float t = red[0][c];
#pragma unroll
    for (int i = 1; i < 8; i++) t += red[i][c];
    t *= alpha;
    const int64_t sg = n / seg;                       // workgroup-uniform (seg % 64 == 0)
    float* o = (sg == 0 ? outs.o[0] : (sg == 1 ? outs.o[1] : outs.o[2])) + (n - sg * seg);
    if (beta != 0.f) t += beta * *o;
    *o = t;
  }
}

int vj_reduce_partials_strided(const float* part, float* out, int64_t P, int64_t N, int64_t stride, float alpha,
                               float beta, hipStream_t stream) {
  if (N == 0) return 0;
  ReduceOuts outs = {{out, nullptr, nullptr}};
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)cdiv64(N, 64)), dim3(512), 0, stream, part, outs,
                     (int64_t)(cdiv64(N, 64) * 64), P, N, stride, alpha, beta);
  VJ_LAUNCH_CHECK("vj_reduce_partials");
  return 0;
}

// part[p][k*D : (k+1)*D] -> outs[k], k < nseg <= 3, in one launch (D % 64 == 0; otherwise one launch per output)
int vj_reduce_partials_multi(const float* part, float* const* outs, int nseg, int64_t P, int64_t D, float alpha, float beta,
                             hipStream_t stream) {
  if (D == 0 || nseg == 0) return 0;
  if (D % 64 != 0) {
    for (int k = 0; k < nseg; k++) {
      int rc = vj_reduce_partials_strided(part + k * D, outs[k], P, D, (int64_t)nseg * D, alpha, beta, stream);
      if (rc) return rc;
    }
    return 0;
  }
  ReduceOuts ro = {{outs[0], nseg > 1 ? outs[1] : nullptr, nseg > 2 ? outs[2] : nullptr}};
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)cdiv64(nseg * D, 64)), dim3(512), 0, stream, part, ro, D, P,
                     (int64_t)nseg * D, (int64_t)nseg * D, alpha, beta);
  VJ_LAUNCH_CHECK("vj_reduce_partials(multi)");
  return 0;
}

extern "C" int vj_reduce_partials(const float* part, float* out, int64_t P, int64_t N, float alpha, float beta,
                                  hipStream_t stream) {
  return vj_reduce_partials_strided(part, out, P, N, N, alpha, beta, stream);
}

// Several independent partial reductions in ONE launch: segment s computes out_s[n] = alpha * sum_p part_s[p * stride_s + n]
// (+ beta * out_s[n]), n < N_s.  Same per-column arithmetic and summation order as reduce_partials_kernel (8 partial lanes,
// fixed-order combine), so a reduction gives the same bits whether it runs alone or as a segment here.  The backward of a
// transformer block ends with one such launch (LayerNorm dgamma | dbeta | proj / fc2 bias sums of both norms + the qkv and fc1
// bias partials of the producing kernels) instead of six reduction / column-sum launches.
#define VJ_REDUCE_MAX_SEGS 16
struct ReduceSegs {
  const float* part[VJ_REDUCE_MAX_SEGS];
  float* out[VJ_REDUCE_MAX_SEGS];
  int64_t P[VJ_REDUCE_MAX_SEGS], N[VJ_REDUCE_MAX_SEGS], stride[VJ_REDUCE_MAX_SEGS];
  int blk_end[VJ_REDUCE_MAX_SEGS];   // exclusive prefix sums of the segments' workgroup counts (cdiv(N, 64) each)
  int n;
};
__global__ __launch_bounds__(512) void reduce_segments_kernel(ReduceSegs rs, float alpha, float beta) {
  __shared__ float red[8][64];
  int sg = 0;
  while (sg + 1 < rs.n && (int)blockIdx.x >= rs.blk_end[sg]) sg++;      // workgroup-uniform
  const int blk0 = sg == 0 ? 0 : rs.blk_end[sg - 1];
  const float* part = rs.part[sg];
  const int64_t P = rs.P[sg], N = rs.N[sg], stride = rs.stride[sg];
  const int c = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int64_t n = (int64_t)((int)blockIdx.x - blk0) * 64 + c;
  float s = 0.f;
  if (n < N) {
#pragma unroll 8
    for (int64_t p = pl; p < P; p += 8) s += part[p * stride + n];
  }
  red[pl][c] = s;
  __syncthreads();
  if (pl == 0 && n < N) {
    float t = red[0][c];
#pragma unroll
    for (int i = 1; i < 8; i++) t += red[i][c];
    t *= alpha;
    float* o = rs.out[sg] + n;
    if (beta != 0.f) t += beta * *o;
    *o = t;
  }
}

extern "C" int vj_reduce_segments(const vj_reduce_seg_t* segs, int64_t n_segs, float alpha, float beta, hipStream_t stream) {
  VJ_CHECK_ARG(segs != nullptr && n_segs >= 0 && n_segs <= VJ_REDUCE_MAX_SEGS, "vj_reduce_segments: 0..%d segments", VJ_REDUCE_MAX_SEGS);
  ReduceSegs rs;
  int nb = 0, k = 0;
  for (int64_t i = 0; i < n_segs; i++) {
    const vj_reduce_seg_t& sg = segs[i];
    VJ_CHECK_ARG(sg.P >= 0 && sg.N >= 0 && sg.stride >= sg.N, "vj_reduce_segments: segment %ld has bad dims", (long)i);
    if (sg.N == 0) continue;
    VJ_CHECK_ARG(sg.out != nullptr && (sg.part != nullptr || sg.P == 0), "vj_reduce_segments: segment %ld has null pointers", (long)i);
    rs.part[k] = sg.part;
    rs.out[k] = sg.out;
    rs.P[k] = sg.P;
    rs.N[k] = sg.N;
    rs.stride[k] = sg.stride;
    nb += (int)cdiv64(sg.N, 64);
    rs.blk_end[k] = nb;
    k++;
  }
  if (k == 0) return 0;
  rs.n = k;
  hipLaunchKernelGGL(reduce_segments_kernel, dim3((unsigned)nb), dim3(512), 0, stream, rs, alpha, beta);
  VJ_LAUNCH_CHECK("vj_reduce_segments");
  return 0;
}

extern "C" int64_t vj_colsum_ws_bytes(int64_t N) { return (int64_t)VJ_COLSUM_PARTS * N * 4; }

extern "C" int vj_colsum_bf16(const void* in, int64_t M, int64_t N, int64_t ld, int64_t group, int64_t row_lo,
                              int64_t row_hi, float* out, float alpha, float beta, void* ws, int64_t ws_bytes,
                              hipStream_t stream) {
  VJ_CHECK_ARG(N % 8 == 0 && ld % 8 == 0, "vj_colsum_bf16: N and ld must be multiples of 8");
  VJ_CHECK_ARG(ws_bytes >= vj_colsum_ws_bytes(N), "vj_colsum_bf16: workspace too small (%ld < %ld)", (long)ws_bytes,
               (long)vj_colsum_ws_bytes(N));
  if (N == 0) return 0;
  if (group <= 0) group = (M > 0 ? M : 1);
  // row chunks: enough workgroups (>= ~2048, 8 per CU) to keep HBM busy when N is narrow (N = 1024: 4 column groups), at
  // least 8 rows per row lane and chunk; the chunk count only changes the (fixed, deterministic) summation order
  const int64_t gx = cdiv64(N, 256);
  int64_t parts = cdiv64(2048, gx);
  if (parts < 64) parts = 64;
  if (parts > VJ_COLSUM_PARTS) parts = VJ_COLSUM_PARTS;
  while (parts > 64 && M / parts < 64) parts /= 2;
  dim3 grid((unsigned)gx, (unsigned)parts);
  hipLaunchKernelGGL(colsum_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)in, (float*)ws, M, N, ld, group,
                     row_lo, row_hi, (int)parts);
  VJ_LAUNCH_CHECK("vj_colsum_bf16");
  return vj_reduce_partials((const float*)ws, out, parts, N, alpha, beta, stream);
}

// ---------------------------------------------------------------------------------------------
// copy_rows_strided: dst[b, dst_off + j, :] = src[b, src_off + j, :] for j < n  (bf16 rows; used to split the
// predictor stream into its context rows (grad of predictor_embed) and to slice target rows).  src == nullptr: the
// destination rows are ZEROED (the context rows of the predictor trunk's output gradient: no ATen fill on the step).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                        int64_t B, int64_t src_rows, int64_t src_off,
                                                        int64_t dst_rows, int64_t dst_off, int64_t n, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < B * n; r += nw) {
    const int64_t b = r / n, j = r - b * n;
    u32x4_t* dp = (u32x4_t*)(dst + (b * dst_rows + dst_off + j) * D);
    if (src != nullptr) {   // (kernel argument: uniform)
      const u32x4_t* sp = (const u32x4_t*)(src + (b * src_rows + src_off + j) * D);
      for (int c = lane; c < D / 8; c += 64) dp[c] = sp[c];
    } else {
      for (int c = lane; c < D / 8; c += 64) dp[c] = (u32x4_t){0u, 0u, 0u, 0u};
    }
  }
}

extern "C" int vj_copy_rows(const void* src, void* dst, int64_t B, int64_t src_rows, int64_t src_off,
                            int64_t dst_rows, int64_t dst_off, int64_t n, int64_t D, hipStream_t stream) {
  VJ_CHECK_ARG(D % 8 == 0, "vj_copy_rows: D must be a multiple of 8");
  VJ_CHECK_ARG((src == nullptr || src_off + n <= src_rows) && dst_off + n <= dst_rows, "vj_copy_rows: slice out of range");
  if (B * n == 0) return 0;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(rows_grid(B * n)), dim3(256), 0, stream, (const bf16_t*)src,
                     (bf16_t*)dst, B, src_rows, src_off, dst_rows, dst_off, n, (int)D);
  VJ_LAUNCH_CHECK("vj_copy_rows");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// transpose_multi: ONE launch that transposes many bf16 matrices (the W^T dgrad shadows of every Linear, refreshed
// once per optimizer step).  desc[t] = {src, dst, M, N, ld_in, Mpad}; blocks[b] = {tensor, tile_m, tile_n, 0}.
// ---------------------------------------------------------------------------------------------
struct TransposeDesc {
  const bf16_t* src;
  bf16_t* dst;
  int64_t M, N, ld_in, Mpad;
};

__global__ __launch_bounds__(256) void transpose_multi_kernel(const TransposeDesc* __restrict__ desc,
                                                              const int4* __restrict__ blocks) {
  __shared__ bf16_t tile[64][66];
  const int4 bi = blocks[blockIdx.x];
  const TransposeDesc d = desc[bi.x];
  const int64_t m0 = (int64_t)bi.y * 64, n0 = (int64_t)bi.z * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int q = t + it * 256;
    const int r = q >> 3, c = (q & 7) * 8;
    u32x4_t v = {0, 0, 0, 0};
    if (m0 + r < d.M && n0 + c < d.N) v = *(const u32x4_t*)(d.src + (m0 + r) * d.ld_in + n0 + c);
#pragma unroll
    for (int i = 0; i < 4; i++) *(uint32_t*)&tile[r][c + 2 * i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int q = t + it * 256;
    const int n = q >> 3, mc = (q & 7) * 8;
    if (n0 + n < d.N && m0 + mc < d.Mpad) {
      u32x4_t o;
#pragma unroll
      for (int i = 0; i < 4; i++)
        o[i] = (uint32_t)tile[mc + 2 * i][n] | ((uint32_t)tile[mc + 2 * i + 1][n] << 16);
      *(u32x4_t*)(d.dst + (n0 + n) * d.Mpad + m0 + mc) = o;
    }
  }
}

// desc: device array of 6 x int64 per tensor {src, dst, M, N, ld_in, Mpad}; blocks: device int32[4*n_blocks]
extern "C" int vj_transpose_multi(const void* desc, const void* blocks, int64_t n_blocks, hipStream_t stream) {
  if (n_blocks == 0) return 0;
  VJ_CHECK_ARG(n_blocks < (1ll << 31), "vj_transpose_multi: too many blocks");
  hipLaunchKernelGGL(transpose_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream,
                     (const TransposeDesc*)desc, (const int4*)blocks);
  VJ_LAUNCH_CHECK("vj_transpose_multi");
  return 0;
}
