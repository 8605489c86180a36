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

// LDS read throughput of one CU with 8 waves issuing back-to-back reads (no MFMA, no global traffic):
// mode 0 = ds_read_b128 (lane-linear, conflict-free), 1 = ds_read_b64_tr_b16 with the TN GEMM's fragment addressing
// (gemm8_tn.hip: 256-byte token rows, chunk ^ ((row & 7) << 1)), 2 = ds_read_b64 (lane-linear).
// out[wave] = cycles for iters x 8 reads; bytes/clk/CU = 8 waves * iters * 8 * 64 lanes * {16, 8, 8} / max cycles.
template <int MODE>
__global__ __launch_bounds__(512) void probe_lds_bw_kernel(long long* out, int iters) {
  extern __shared__ char probe_lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 128 * 1024 / 4; i += 512) ((uint32_t*)probe_lds)[i] = i;
  __syncthreads();
  unsigned addr;
  if (MODE == 1) {
    const int g = lane >> 4, jj = (lane & 15) >> 2, q4 = lane & 3;
    const int row_l = 4 * g + jj, key2 = (row_l & 7) << 1;
    addr = (unsigned)(uintptr_t)probe_lds + w * 16384 + row_l * 256 + (q4 & 1) * 8 + ((((w & 3) * 4) ^ key2) | (q4 >> 1)) * 16;
  } else {
    addr = (unsigned)(uintptr_t)probe_lds + w * 16384 + lane * (MODE == 0 ? 16 : 8);
  }
  uint32_t sink = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
      u32x4_t r[8];
#pragma unroll
      for (int k = 0; k < 8; k++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k]) : "v"(addr), "n"(1024 * 0) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; k++) sink ^= r[k][0];
    } else {
      u32x2_t r[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (MODE == 1) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[k]) : "v"(addr) : "memory");
        else asm volatile("ds_read_b64 %0, %1" : "=v"(r[k]) : "v"(addr) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; k++) sink ^= r[k][0];
    }
  }
  const long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
  if (sink == 0x12345678u) out[0] = 0;
}
extern "C" int vj_probe_lds_bw(long long* out, int mode, int iters, int n_wgs, hipStream_t stream) {
  VJ_CHECK_ARG(mode >= 0 && mode <= 2 && iters > 0 && n_wgs > 0, "vj_probe_lds_bw: bad arguments");
  static VjPerDeviceOnce attr_once;   // the dynamic-LDS limit is a per-device attribute of the function
  attr_once([] {
    (void)hipFuncSetAttribute((const void*)probe_lds_bw_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)probe_lds_bw_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void*)probe_lds_bw_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  });
  if (mode == 0) hipLaunchKernelGGL(probe_lds_bw_kernel<0>, dim3(n_wgs), dim3(512), 128 * 1024, stream, out, iters);
  else if (mode == 1) hipLaunchKernelGGL(probe_lds_bw_kernel<1>, dim3(n_wgs), dim3(512), 128 * 1024, stream, out, iters);
  else hipLaunchKernelGGL(probe_lds_bw_kernel<2>, dim3(n_wgs), dim3(512), 128 * 1024, stream, out, iters);
  VJ_LAUNCH_CHECK("vj_probe_lds_bw");
  return 0;
}

// One wave that does nothing for `ticks` periods of the constant 100 MHz timer.  Two of them launched on two streams finish in one
// spin time when the streams run concurrently and in two when they are serialised -- i.e. when the runtime mapped both streams
// to the SAME hardware queue (ROCclr hands its GPU_MAX_HW_QUEUES queues to streams round-robin).  The engine uses it once per
// process to choose side / update / communication streams that are independent of each other (engine/layers.py independent_stream):
// a side stream that shares the main stream's queue turns the two-stream step into a serial one (+19 % measured, round 5).
__global__ void probe_spin_kernel(long long ticks, long long* stamps) {
  const long long t0 = wall_clock64();
  long long t = t0;
  while (t - t0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    t = wall_clock64();
  }
  if (stamps != nullptr && threadIdx.x == 0) {   // {start, end} on the chip-wide 100 MHz timer
    stamps[0] = t0;
    stamps[1] = t;
  }
}
extern "C" int vj_probe_spin(int64_t ticks, hipStream_t stream) {
  VJ_CHECK_ARG(ticks >= 0 && ticks <= 100000000, "vj_probe_spin: ticks (100 MHz) outside [0, 1e8]");
  hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(64), 0, stream, (long long)ticks, (long long*)nullptr);
  VJ_LAUNCH_CHECK("vj_probe_spin");
  return 0;
}
// The same kernel leaving its start and end stamps (the chip-wide constant 100 MHz timer, identical on every CU) in stamps[0..1]
// (device memory): two of them on two streams OVERLAP on the device's own clock iff the streams sit on different hardware queues --
// the host's wall clock, which 8 ranks and their data-loader workers perturb, is not involved (round-5 advisor finding).
extern "C" int vj_probe_spin_stamped(int64_t ticks, int64_t* stamps, hipStream_t stream) {
  VJ_CHECK_ARG(ticks >= 0 && ticks <= 100000000, "vj_probe_spin_stamped: ticks (100 MHz) outside [0, 1e8]");
  VJ_CHECK_ARG(stamps != nullptr, "vj_probe_spin_stamped: null stamps");
  hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(64), 0, stream, (long long)ticks, (long long*)stamps);
  VJ_LAUNCH_CHECK("vj_probe_spin_stamped");
  return 0;
}
