// How many vector / LDS instructions hide in the gap behind one MFMA of the SAME wave on a gfx950 SIMD, with ONE or TWO waves per
// SIMD?  (round 6: the round-3 probe, valu_probe.hip, measured 2-4 waves per SIMD only and found that the pipes add; MI355X_MICROARCH.md's
// one-wave-per-SIMD attention numbers say 5 single-issue fillers hide under a 32-cycle 32x32x16 MFMA.  The fused attention backward is
// designed on the answer.)
//
// One workgroup of 4 x W waves on one CU.  Every wave runs a counted loop of 16 MFMAs (4 independent accumulators in rotation, so that no
// MFMA waits for its own accumulator), each followed by K fillers of one kind on independent registers; everything is inline assembly,
// so program order = issue order.  s_memtime around the loop; cycles per MFMA group are printed for K = 0..6.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_filler_probe.hip -o gpurun_out/mfma_filler_probe && gpurun_out/mfma_filler_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) unsigned u2;

#define REP 1000

enum { F_FMA = 0, F_EXP, F_CVT, F_MUL, F_TR, F_MIX, F_WR, F_NKIND };
static const char* kind_name[] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "v_mul_f32", "ds_read_b64_tr_b16", "mix exp,mul,cvt,tr", "ds_write_b64"};

template <int KIND>
__device__ __forceinline__ void filler(int n, float (&r)[8], u2 (&lt)[8], float a, float b, unsigned laddr) {
  const int i = n & 7;
  int k = KIND;
  if (KIND == F_MIX) k = (n & 3) == 0 ? F_EXP : ((n & 3) == 1 ? F_MUL : ((n & 3) == 2 ? F_CVT : F_TR));
  if (k == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
  if (k == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
  if (k == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
  if (k == F_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
  if (k == F_TR) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lt[i]) : "v"(laddr));
  if (k == F_WR) asm volatile("ds_write_b64 %0, %1" ::"v"(laddr), "v"(lt[i]));
}

// SHAPE 0: v_mfma_f32_16x16x32_bf16 (4 passes), 1: v_mfma_f32_32x32x16_bf16 (8 passes)
template <int SHAPE, int KIND, int K>
__global__ void probe(long long* out, float* sink) {
  float r[8];
  u2 lt[8];
  const float a = 1.0001f, b = 0.5f;
  for (int i = 0; i < 8; i++) {
    r[i] = 0.001f * (threadIdx.x + i);
    lt[i] = (u2){threadIdx.x, (unsigned)i};
  }
  __shared__ __attribute__((aligned(16))) char lbuf[32768];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((unsigned*)lbuf)[i] = i;
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lbuf + (threadIdx.x & 511) * 8;
  bf16x8_t x, y;
  for (int i = 0; i < 8; i++) {
    x[i] = (__bf16)(0.01f * i);
    y[i] = (__bf16)(0.02f * i);
  }
  f32x4_t c4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16_t c16[4];
  for (int j = 0; j < 4; j++)
    for (int e = 0; e < 16; e++) c16[j][e] = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < REP; it++) {
#pragma unroll
    for (int m = 0; m < 16; m++) {
      if constexpr (SHAPE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4[m & 3]) : "v"(x), "v"(y));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c16[m & 3]) : "v"(x), "v"(y));
#pragma unroll
      for (int f = 0; f < K; f++) filler<KIND>(m * K + f, r, lt, a, b, laddr);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += r[i] + (float)lt[i][0];
  for (int j = 0; j < 4; j++) s += c4[j][0] + c16[j][3];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = (long long)(t1 - t0);
}

template <int SHAPE, int KIND, int K>
static double run(int waves_per_simd, long long* d_out, float* d_sink) {
  const int nw = 4 * waves_per_simd;
  probe<SHAPE, KIND, K><<<1, 64 * nw>>>(d_out, d_sink);
  hipDeviceSynchronize();
  probe<SHAPE, KIND, K><<<1, 64 * nw>>>(d_out, d_sink);
  hipDeviceSynchronize();
  std::vector<long long> h(nw);
  hipMemcpy(h.data(), d_out, nw * sizeof(long long), hipMemcpyDeviceToHost);
  long long mx = 0;
  for (auto v : h) mx = v > mx ? v : mx;
  return (double)mx / (REP * 16.0);
}

template <int SHAPE, int KIND>
static void row(long long* d_out, float* d_sink) {
  for (int w = 1; w <= 2; w++) {
    double c[7];
    c[0] = run<SHAPE, KIND, 0>(w, d_out, d_sink);
    c[1] = run<SHAPE, KIND, 1>(w, d_out, d_sink);
    c[2] = run<SHAPE, KIND, 2>(w, d_out, d_sink);
    c[3] = run<SHAPE, KIND, 3>(w, d_out, d_sink);
    c[4] = run<SHAPE, KIND, 4>(w, d_out, d_sink);
    c[5] = run<SHAPE, KIND, 5>(w, d_out, d_sink);
    c[6] = run<SHAPE, KIND, 6>(w, d_out, d_sink);
    printf("| %s | %s | %d |", SHAPE == 0 ? "16x16x32" : "32x32x16", kind_name[KIND], w);
    for (int k = 0; k < 7; k++) printf(" %.1f |", c[k] * w);   // x w: cycles of the SIMD per MFMA group (w waves share it)
    printf("\n");
  }
}

int main() {
  long long* d_out;
  float* d_sink;
  hipMalloc(&d_out, 64 * sizeof(long long));
  hipMalloc(&d_sink, 64);
  printf("SIMD cycles per {1 MFMA + K fillers} (s_memtime ticks of the slowest wave / groups, x waves per SIMD)\n");
  printf("| MFMA | filler | waves/SIMD | K=0 | 1 | 2 | 3 | 4 | 5 | 6 |\n|---|---|---|---|---|---|---|---|---|---|\n");
  row<0, F_FMA>(d_out, d_sink);
  row<0, F_EXP>(d_out, d_sink);
  row<0, F_CVT>(d_out, d_sink);
  row<0, F_MUL>(d_out, d_sink);
  row<0, F_TR>(d_out, d_sink);
  row<0, F_WR>(d_out, d_sink);
  row<0, F_MIX>(d_out, d_sink);
  row<1, F_FMA>(d_out, d_sink);
  row<1, F_EXP>(d_out, d_sink);
  row<1, F_CVT>(d_out, d_sink);
  row<1, F_TR>(d_out, d_sink);
  row<1, F_MIX>(d_out, d_sink);
  return 0;
}
