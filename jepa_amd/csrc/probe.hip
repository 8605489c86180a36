// Hardware probes used by tests/profiles to pin down gfx950 behaviour the kernels rely on (or will rely on).
#include "common.hpp"

// ds_read_b64_tr_b16 lane mapping: LDS holds u16 value i at element i; lane l reads at byte address
// base + addr_scale*l; out[l*4 + j] = element index delivered to lane l, slot j.
__global__ void probe_tr16_kernel(uint32_t* out, int addr_scale) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const unsigned addr = (unsigned)(uintptr_t)lds + threadIdx.x * addr_scale;
  u32x2_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = r[0] & 0xffffu;
  out[threadIdx.x * 4 + 1] = r[0] >> 16;
  out[threadIdx.x * 4 + 2] = r[1] & 0xffffu;
  out[threadIdx.x * 4 + 3] = r[1] >> 16;
}

extern "C" int vj_probe_tr16(uint32_t* out256, int addr_scale, hipStream_t stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, stream, out256, addr_scale);
  VJ_LAUNCH_CHECK("vj_probe_tr16");
  return 0;
}

// streaming copy, 16 B per lane: the achievable-HBM yardstick the roofline fractions are read against
__global__ __launch_bounds__(256) void probe_copy_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst,
                                                         int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int vj_probe_copy(const void* src, void* dst, int64_t bytes, hipStream_t stream) {
  VJ_CHECK_ARG(bytes % 16 == 0, "vj_probe_copy: bytes must be a multiple of 16");
  hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 8), dim3(256), 0, stream, (const u32x4_t*)src, (u32x4_t*)dst,
                     bytes / 16);
  VJ_LAUNCH_CHECK("vj_probe_copy");
  return 0;
}
