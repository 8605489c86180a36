// Joules per FLOP of the two bf16 MFMA shapes of gfx950 at the package power cap (round 6, verdict item 6).
//
// The step's GEMM loops run at the 1400 W cap (profiles/r05_power_matrix.md): what lowers the energy of a FLOP raises the clock.
// v_mfma_f32_32x32x16_bf16 reads half the A / B operand registers per FLOP of v_mfma_f32_16x16x32_bf16 but moves twice the accumulator
// registers.  This probe runs both on the SAME 64 x 64 x 32 block-step per wave (16 x 4-pass 16x16x32 or 8 x 8-pass
// 32x32x16: the same FLOPs, the same 32 operand VGPRs, 64 accumulator VGPRs), every CU filled with one or two
// waves per SIMD, operands either held in registers or re-read from LDS every block-step with ds_read_b128 as a GEMM K loop does,
// with random or all-zero data.  The host side (tools/mfma_power.py) samples package power and shader clock while it loops.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/mfma_power_probe.hip -o /tmp/libmfma_power_probe.so
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int SHAPE>
__device__ __forceinline__ void block_step(const bf16x8_t (&a)[4], const bf16x8_t (&b)[4], f32x4_t (&c4)[16], f32x16_t (&c16)[4]) {
  if constexpr (SHAPE == 0) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4[i * 4 + j]) : "v"(a[i]), "v"(b[j]));
  } else {
    // a[2 * rb + kh]: row block rb (32 rows), k half kh (16 of the 32 k); the same for b
#pragma unroll
    for (int kh = 0; kh < 2; kh++)
#pragma unroll
      for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c16[rb * 2 + cb]) : "v"(a[2 * rb + kh]), "v"(b[2 * cb + kh]));
  }
}

template <int SHAPE, int LDS>
__global__ __launch_bounds__(512) void mfma_power_kernel(const bf16x8_t* in, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) bf16x8_t lbuf[LDS ? 512 * 16 : 1];   // 128 KB: one workgroup per CU
  const bf16x8_t* p = in + ((size_t)blockIdx.x * 512 + threadIdx.x) * 16;
  bf16x8_t a[2][4], b[2][4];
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      a[s][i] = p[s * 8 + i];
      b[s][i] = p[s * 8 + 4 + i];
    }
  unsigned laddr = 0;
  if constexpr (LDS) {
    // fragment f of all lanes contiguous: lane stride 16 B (conflict-free ds_read_b128)
#pragma unroll
    for (int f = 0; f < 16; f++) lbuf[f * 512 + threadIdx.x] = p[f];
    laddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lbuf + threadIdx.x * 16;
    __syncthreads();
  }
  f32x4_t c4[16];
  f32x16_t c16[4];
#pragma unroll
  for (int j = 0; j < 16; j++) c4[j] = (f32x4_t){0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int e = 0; e < 16; e++) c16[j][e] = 0.f;
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      if constexpr (LDS) {
        // the operands of the other set for the next block-step are requested before this step's MFMAs, as a K loop does
#pragma unroll
        for (int i = 0; i < 4; i++) {
          // fragment f sits at f * 8192 bytes; the offset field holds 16 bits, hence one base per operand set
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[s ^ 1][i]) : "v"(laddr + (s ^ 1) * 65536), "n"(i * 8192) : "memory");
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[s ^ 1][i]) : "v"(laddr + (s ^ 1) * 65536), "n"((4 + i) * 8192) : "memory");
        }
      }
      block_step<SHAPE>(a[s], b[s], c4, c16);
      if constexpr (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; j++) s += c4[j][0] + c4[j][3];
#pragma unroll
  for (int j = 0; j < 4; j++) s += c16[j][0] + c16[j][7];
  if (s == 12345.678f) sink[0] = s;
}

extern "C" int mfma_power_launch(int shape, int lds, int waves_per_cu, int cus, int iters, const void* in, float* sink, hipStream_t stream) {
  const dim3 grid(cus), block(64 * waves_per_cu);
  const bf16x8_t* p = (const bf16x8_t*)in;
  if (shape == 0 && lds == 0) hipLaunchKernelGGL((mfma_power_kernel<0, 0>), grid, block, 0, stream, p, sink, iters);
  else if (shape == 0) hipLaunchKernelGGL((mfma_power_kernel<0, 1>), grid, block, 0, stream, p, sink, iters);
  else if (lds == 0) hipLaunchKernelGGL((mfma_power_kernel<1, 0>), grid, block, 0, stream, p, sink, iters);
  else hipLaunchKernelGGL((mfma_power_kernel<1, 1>), grid, block, 0, stream, p, sink, iters);
  return (int)hipGetLastError();
}
