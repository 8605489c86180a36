// Shared pieces of the bf16 MFMA GEMM kernels: argument block, epilogue selector and the fused epilogue.
#pragma once
#include "common.hpp"

enum { EPI_BF16 = 0, EPI_GELU = 1, EPI_DGELU = 2, EPI_F32 = 3 };

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;      // [N] fp32, nullable
  const bf16_t* res;      // [M,N] bf16 residual, nullable (EPI_BF16)
  const bf16_t* aux_in;   // [M,N] bf16 pre-activation u (EPI_DGELU)
  bf16_t* aux_out;        // [M,N] bf16 pre-activation u out, nullable (EPI_GELU)
  int64_t M, N, K, lda, ldb, ldc, ldr, ldaux;
  float alpha, beta;
  int tiles_m, tiles_n;
  int splitk;          // > 1: K is cut into `splitk` slices, raw fp32 partials go to ws[slice][M][N] (EPI_F32 only)
  int ktiles_per;      // K-tiles per slice
  float* ws;
};


// Fused epilogue.  acc[i][j] comes from MFMA 16x16x32 issued with swapped operands (D = Bfrag x Afrag): lane
// (g = lane>>4, r = lane&15) owns row m = .. + i*16 + r and the 4 consecutive columns n = .. + j*16 + 4g + {0..3}.
template <int EPI, int FM, int FN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4_t (&acc)[FM][FN], int64_t m_base,
                                              int64_t n_base, int frow, int fg, int slice) {
  const int64_t ncol0 = n_base + fg * 4;   // this lane's first column; tile j adds j*16
  float4 bias4[FN];
  if constexpr (EPI != EPI_F32) {
#pragma unroll
    for (int j = 0; j < FN; j++) {
      const int64_t n = ncol0 + j * 16;
      bias4[j] = (p.bias && n < p.N) ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int i = 0; i < FM; i++) {
    const int64_t m = m_base + i * 16 + frow;
    if (m >= p.M) continue;
    // row base pointers, computed once per row
    float* c32 = (float*)p.C + m * p.ldc + ncol0;
    bf16_t* c16 = (bf16_t*)p.C + m * p.ldc + ncol0;
    float* wsp = p.ws + ((int64_t)slice * p.M + m) * p.N + ncol0;
    const bf16_t* resp = p.res ? p.res + m * p.ldr + ncol0 : nullptr;
    const bf16_t* auxi = p.aux_in ? p.aux_in + m * p.ldaux + ncol0 : nullptr;
    bf16_t* auxo = p.aux_out ? p.aux_out + m * p.ldaux + ncol0 : nullptr;
#pragma unroll
    for (int j = 0; j < FN; j++) {
      if (ncol0 + j * 16 >= p.N) continue;  // N % 4 == 0 is enforced by the host wrapper
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if constexpr (EPI == EPI_F32) {
        if (p.splitk > 1) {  // raw partial; alpha/beta are applied by the slice reduction
          *(float4*)(wsp + j * 16) = make_float4(v[0], v[1], v[2], v[3]);
          continue;
        }
        float4 o = make_float4(v[0] * p.alpha, v[1] * p.alpha, v[2] * p.alpha, v[3] * p.alpha);
        if (p.beta != 0.f) {
          const float4 c0 = *(const float4*)(c32 + j * 16);
          o.x += p.beta * c0.x;
          o.y += p.beta * c0.y;
          o.z += p.beta * c0.z;
          o.w += p.beta * c0.w;
        }
        *(float4*)(c32 + j * 16) = o;
      } else {
        v[0] += bias4[j].x;
        v[1] += bias4[j].y;
        v[2] += bias4[j].z;
        v[3] += bias4[j].w;
        if constexpr (EPI == EPI_GELU) {
          u32x2_t u;
          u[0] = pack_bf2(v[0], v[1]);
          u[1] = pack_bf2(v[2], v[3]);
          if (auxo) *(u32x2_t*)(auxo + j * 16) = u;
          // GELU of the bf16-rounded pre-activation: the backward pass re-derives gelu'(u) from the same bits
          v[0] = gelu_f(bf_lo(u[0]));
          v[1] = gelu_f(bf_hi(u[0]));
          v[2] = gelu_f(bf_lo(u[1]));
          v[3] = gelu_f(bf_hi(u[1]));
        } else if constexpr (EPI == EPI_DGELU) {
          const u32x2_t u = *(const u32x2_t*)(auxi + j * 16);
          v[0] *= dgelu_f(bf_lo(u[0]));
          v[1] *= dgelu_f(bf_hi(u[0]));
          v[2] *= dgelu_f(bf_lo(u[1]));
          v[3] *= dgelu_f(bf_hi(u[1]));
        } else {
          if (resp) {
            const u32x2_t r2 = *(const u32x2_t*)(resp + j * 16);
            v[0] += bf_lo(r2[0]);
            v[1] += bf_hi(r2[0]);
            v[2] += bf_lo(r2[1]);
            v[3] += bf_hi(r2[1]);
          }
        }
        u32x2_t o;
        o[0] = pack_bf2(v[0], v[1]);
        o[1] = pack_bf2(v[2], v[3]);
        *(u32x2_t*)(c16 + j * 16) = o;
      }
    }
  }
}

// XCD-aware, grouped tile mapping (bijective for any grid size): workgroup `bid` of `nblk` -> logical tile index such
// that each XCD (private L2; hardware dispatches workgroup b to XCD b % 8) works on a contiguous band of tiles.
__device__ __forceinline__ int xcd_logical(int bid, int nblk) {
  const int qx = nblk >> 3, rx = nblk & 7, xcd = bid & 7, pos = bid >> 3;
  return (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + pos;
}
// GM row-tiles form a group that sweeps all column tiles (keeps the A panel hot in L2)
__device__ __forceinline__ void tile_of(int logical, int tiles_m, int tiles_n, int& tm, int& tn) {
  constexpr int GM = 8;
  const int per_group = GM * tiles_n;
  const int group = logical / per_group, in_g = logical - group * per_group;
  const int first_m = group * GM;
  const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}
