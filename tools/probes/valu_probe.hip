// Instruction-cost probe for gfx950 (one workgroup, W waves per SIMD): shader-clock cycles per wave-instruction for the
// vector operations the attention / epilogue code is made of, alone and next to MFMAs.  Every body is inline assembly on
// independent registers (no dependency stalls inside a group), repeated REP times in a counted loop.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/valu_probe.hip -o gpurun_out/valu_probe && gpurun_out/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define REP 2000

#define BODY8(op)  op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)

#define OP_FMA(i)   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(q2));
#define OP_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q2));
#define OP_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q2));
#define OP_EXP(i)   asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define OP_RCP(i)   asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define OP_CVT(i)   asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_MAX3(i)  asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_MOV(i)   asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(a));
#define OP_SWAP(i)  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 1) & 7]));
#define OP_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(r[i]) : "v"(idx));

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void probe(long long* out, float* sink) {
  float r[8];
  f2 p[8];
  const float a = 1.0001f, b = 0.5f;
  const f2 q2 = {1.0001f, 0.9999f};
  const int idx = (threadIdx.x ^ 16) * 4;
  for (int i = 0; i < 8; i++) {
    r[i] = 0.001f * (threadIdx.x + i);
    p[i] = (f2){r[i], r[i] + 1.f};
  }
  __shared__ __attribute__((aligned(16))) char lbuf[16384];
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
  u4 lq[8];
  u2 lt[8];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((unsigned*)lbuf)[i] = i;
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lbuf + (threadIdx.x & 63) * 16;
  f32x4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  bf16x8_t x, y;
  for (int i = 0; i < 8; i++) { x[i] = (__bf16)(0.01f * i); y[i] = (__bf16)(0.02f * i); }
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < REP; it++) {
    if constexpr (KIND == 0) { BODY8(OP_FMA) BODY8(OP_FMA) }
    if constexpr (KIND == 1) { BODY8(OP_PKFMA) BODY8(OP_PKFMA) }
    if constexpr (KIND == 2) { BODY8(OP_EXP) BODY8(OP_EXP) }
    if constexpr (KIND == 3) { BODY8(OP_CVT) BODY8(OP_CVT) }
    if constexpr (KIND == 4) { BODY8(OP_MAX3) BODY8(OP_MAX3) }
    if constexpr (KIND == 5) { BODY8(OP_PKMUL) BODY8(OP_PKADD) }
    if constexpr (KIND == 6) { BODY8(OP_RCP) BODY8(OP_RCP) }
    if constexpr (KIND == 7) { BODY8(OP_SWAP) BODY8(OP_SWAP) }
    if constexpr (KIND == 8) { BODY8(OP_BPERM) BODY8(OP_BPERM) }
    if constexpr (KIND == 9) {   // 16 MFMAs, 4 independent accumulators
      for (int k = 0; k < 4; k++)
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[j], 0, 0, 0);
    }
    if constexpr (KIND == 10) {  // 16 MFMAs, each followed by 3 FMAs (12 cycles of vector issue under a 16-cycle MFMA?)
      for (int k = 0; k < 4; k++)
        for (int j = 0; j < 4; j++) {
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[j], 0, 0, 0);
          OP_FMA(0) OP_FMA(1) OP_FMA(2)
        }
    }
    if constexpr (KIND == 11) {  // 16 MFMAs, each followed by one v_exp + one FMA
      for (int k = 0; k < 4; k++)
        for (int j = 0; j < 4; j++) {
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[j], 0, 0, 0);
          OP_EXP(0) OP_FMA(1)
        }
    }
    if constexpr (KIND == 12) {  // 16 MFMAs then 48 FMAs (not interleaved in program order)
      for (int k = 0; k < 4; k++)
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[j], 0, 0, 0);
      BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA)
    }
    if constexpr (KIND == 13) { BODY8(OP_MOV) BODY8(OP_MOV) }
    // the instruction mix of ONE 64-key tile of the hd = 64 attention forward (32 queries per wave): 32 MFMAs,
    // 32 v_exp, 16 v_pk_fma, 16 v_pk_add, 8 v_pk_mul, 16 v_cvt_pk, 16 v_max3, 16 v_fma (misc)
    if constexpr (KIND == 14) {   // blocked like the kernel: 16 MFMAs | all the vector work | 16 MFMAs
      for (int k = 0; k < 4; k++)
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[j], 0, 0, 0);
      BODY8(OP_MAX3) BODY8(OP_MAX3) BODY8(OP_PKFMA) BODY8(OP_PKFMA) BODY8(OP_EXP) BODY8(OP_EXP) BODY8(OP_EXP) BODY8(OP_EXP)
      BODY8(OP_PKADD) BODY8(OP_PKADD) BODY8(OP_PKMUL) BODY8(OP_CVT) BODY8(OP_CVT) BODY8(OP_FMA) BODY8(OP_FMA)
      for (int k = 0; k < 4; k++)
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[j], 0, 0, 0);
    }
    if constexpr (KIND == 15) {   // the same mix, one MFMA every ~4.4 vector instructions
      for (int k = 0; k < 8; k++) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[0], 0, 0, 0);
        OP_MAX3(0) OP_MAX3(1) OP_PKFMA(0) OP_PKFMA(1) OP_EXP(2)
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[1], 0, 0, 0);
        OP_EXP(3) OP_EXP(4) OP_EXP(5) OP_PKADD(2) OP_PKADD(3)
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[2], 0, 0, 0);
        OP_PKMUL(4) OP_CVT(6) OP_CVT(7) OP_FMA(0) OP_FMA(1)
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[3], 0, 0, 0);
      }
    }
    // one MFMA followed by ~9-13 cycles of ONE kind of vector instruction: which kinds hide under the MFMA?
#define MF(j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[j], 0, 0, 0);
    if constexpr (KIND == 20) { for (int k = 0; k < 4; k++) { MF(0) OP_FMA(0) OP_FMA(1) OP_FMA(2) OP_FMA(3) MF(1) OP_FMA(4) OP_FMA(5) OP_FMA(6) OP_FMA(7) MF(2) OP_FMA(0) OP_FMA(1) OP_FMA(2) OP_FMA(3) MF(3) OP_FMA(4) OP_FMA(5) OP_FMA(6) OP_FMA(7) } }
    if constexpr (KIND == 21) { for (int k = 0; k < 4; k++) { MF(0) OP_PKFMA(0) OP_PKFMA(1) MF(1) OP_PKFMA(2) OP_PKFMA(3) MF(2) OP_PKFMA(4) OP_PKFMA(5) MF(3) OP_PKFMA(6) OP_PKFMA(7) } }
    if constexpr (KIND == 22) { for (int k = 0; k < 4; k++) { MF(0) OP_EXP(0) MF(1) OP_EXP(1) MF(2) OP_EXP(2) MF(3) OP_EXP(3) } }
    if constexpr (KIND == 23) { for (int k = 0; k < 4; k++) { MF(0) OP_CVT(0) OP_CVT(1) MF(1) OP_CVT(2) OP_CVT(3) MF(2) OP_CVT(4) OP_CVT(5) MF(3) OP_CVT(6) OP_CVT(7) } }
    if constexpr (KIND == 24) { for (int k = 0; k < 4; k++) { MF(0) OP_MAX3(0) OP_MAX3(1) MF(1) OP_MAX3(2) OP_MAX3(3) MF(2) OP_MAX3(4) OP_MAX3(5) MF(3) OP_MAX3(6) OP_MAX3(7) } }
    if constexpr (KIND == 25) { for (int k = 0; k < 4; k++) { MF(0) OP_EXP(0) OP_EXP(1) MF(1) OP_EXP(2) OP_EXP(3) MF(2) OP_EXP(4) OP_EXP(5) MF(3) OP_EXP(6) OP_EXP(7) } }
    if constexpr (KIND == 26) { for (int k = 0; k < 4; k++) { MF(0) OP_PKFMA(0) OP_PKFMA(1) OP_PKFMA(2) OP_PKFMA(3) MF(1) OP_PKFMA(4) OP_PKFMA(5) OP_PKFMA(6) OP_PKFMA(7) MF(2) OP_PKFMA(0) OP_PKFMA(1) OP_PKFMA(2) OP_PKFMA(3) MF(3) OP_PKFMA(4) OP_PKFMA(5) OP_PKFMA(6) OP_PKFMA(7) } }
    // LDS reads: issue cost alone and next to MFMAs (results land in scratch registers, one lgkmcnt(0) per group)
#define OP_LDS(i)  asm volatile("ds_read_b128 %0, %1 offset:" #i "*1024" : "=v"(lq[i]) : "v"(laddr));
#define OP_LDST(i) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #i "*1024" : "=v"(lt[i]) : "v"(laddr));
    if constexpr (KIND == 30) { BODY8(OP_LDS) BODY8(OP_LDS) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (KIND == 31) { BODY8(OP_LDST) BODY8(OP_LDST) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (KIND == 32) { for (int k = 0; k < 4; k++) { MF(0) OP_LDS(0) MF(1) OP_LDS(1) MF(2) OP_LDS(2) MF(3) OP_LDS(3) } asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (KIND == 33) { for (int k = 0; k < 4; k++) { MF(0) OP_LDS(0) OP_LDS(1) MF(1) OP_LDS(2) OP_LDS(3) MF(2) OP_LDS(4) OP_LDS(5) MF(3) OP_LDS(6) OP_LDS(7) } asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (KIND == 34) { for (int k = 0; k < 4; k++) { MF(0) OP_LDST(0) OP_LDST(1) MF(1) OP_LDST(2) OP_LDST(3) MF(2) OP_LDST(4) OP_LDST(5) MF(3) OP_LDST(6) OP_LDST(7) } asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (KIND == 35) {   // barrier cost: 16 MFMAs, one s_barrier
      for (int k = 0; k < 4; k++) { MF(0) MF(1) MF(2) MF(3) }
      __builtin_amdgcn_s_barrier();
    }
    if constexpr (KIND == 36) {   // 16 MFMAs in 4 groups with a barrier after each
      for (int k = 0; k < 4; k++) { MF(0) MF(1) MF(2) MF(3) __builtin_amdgcn_s_barrier(); }
    }
    // the tile mix with every packed f32 instruction replaced by two plain ones
    if constexpr (KIND == 17) {
      for (int k = 0; k < 4; k++) { MF(0) MF(1) MF(2) MF(3) }
      BODY8(OP_MAX3) BODY8(OP_MAX3) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_EXP) BODY8(OP_EXP) BODY8(OP_EXP) BODY8(OP_EXP)
      BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_CVT) BODY8(OP_CVT) BODY8(OP_FMA) BODY8(OP_FMA)
      for (int k = 0; k < 4; k++) { MF(0) MF(1) MF(2) MF(3) }
    }
    if constexpr (KIND == 18) {
      BODY8(OP_MAX3) BODY8(OP_MAX3) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_EXP) BODY8(OP_EXP) BODY8(OP_EXP) BODY8(OP_EXP)
      BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_FMA) BODY8(OP_CVT) BODY8(OP_CVT) BODY8(OP_FMA) BODY8(OP_FMA)
    }
    if constexpr (KIND == 16) {   // vector part of the tile alone
      BODY8(OP_MAX3) BODY8(OP_MAX3) BODY8(OP_PKFMA) BODY8(OP_PKFMA) BODY8(OP_EXP) BODY8(OP_EXP) BODY8(OP_EXP) BODY8(OP_EXP)
      BODY8(OP_PKADD) BODY8(OP_PKADD) BODY8(OP_PKMUL) BODY8(OP_CVT) BODY8(OP_CVT) BODY8(OP_FMA) BODY8(OP_FMA)
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += r[i] + p[i][0] + p[i][1];
  for (int j = 0; j < 4; j++) s += acc[j][0];
  for (int i = 0; i < 8; i++) s += (float)(lq[i][0] + lt[i][0]);
  if (s == 123.456f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

struct Case { const char* name; int kind; int per_iter; };

template <int K>
static void run(const char* name, int per_iter, int waves_per_simd, long long* dout, float* sink) {
  const int threads = 256 * waves_per_simd;   // 4 SIMDs x waves_per_simd
  hipLaunchKernelGGL(probe<K>, dim3(1), dim3(threads), 0, 0, dout, sink);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(probe<K>, dim3(1), dim3(threads), 0, 0, dout, sink);
  hipDeviceSynchronize();
  std::vector<long long> h(threads / 64);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  long long mx = 0;
  for (auto v : h) mx = v > mx ? v : mx;
  // clock64() = s_memtime counts at a fixed 100 MHz on gfx9; report both raw ticks and per-group figures
  printf("%-44s waves/SIMD %d: %8lld ticks per %d iterations -> %.3f ticks per group of %d instr (x waves %d)\n", name,
         waves_per_simd, mx, REP, (double)mx / REP, per_iter, waves_per_simd);
}

int main() {
  long long* dout;
  float* sink;
  hipMalloc(&dout, 4096);
  hipMalloc(&sink, 64);
  for (int w = 2; w <= 4; w += 2) {
    run<13>("16 x v_mov_b32", 16, w, dout, sink);
    run<0>("16 x v_fma_f32", 16, w, dout, sink);
    run<1>("16 x v_pk_fma_f32", 16, w, dout, sink);
    run<5>("8 x v_pk_mul_f32 + 8 x v_pk_add_f32", 16, w, dout, sink);
    run<2>("16 x v_exp_f32", 16, w, dout, sink);
    run<6>("16 x v_rcp_f32", 16, w, dout, sink);
    run<3>("16 x v_cvt_pk_bf16_f32", 16, w, dout, sink);
    run<4>("16 x v_max3_f32", 16, w, dout, sink);
    run<7>("16 x v_permlane32_swap_b32", 16, w, dout, sink);
    run<8>("16 x (ds_bpermute_b32 + wait)", 16, w, dout, sink);
    run<9>("16 x mfma_16x16x32_bf16", 16, w, dout, sink);
    run<10>("16 x (mfma + 3 v_fma_f32)", 64, w, dout, sink);
    run<11>("16 x (mfma + v_exp_f32 + v_fma_f32)", 48, w, dout, sink);
    run<12>("16 x mfma, then 48 x v_fma_f32", 64, w, dout, sink);
    run<20>("16 x (mfma + 4 v_fma_f32)   [alone 16+10.6]", 80, w, dout, sink);
    run<21>("16 x (mfma + 2 v_pk_fma_f32) [alone 16+8.7]", 48, w, dout, sink);
    run<26>("16 x (mfma + 4 v_pk_fma_f32) [alone 16+17.4]", 80, w, dout, sink);
    run<22>("16 x (mfma + 1 v_exp_f32)   [alone 16+8.2]", 32, w, dout, sink);
    run<25>("16 x (mfma + 2 v_exp_f32)   [alone 16+16.4]", 48, w, dout, sink);
    run<23>("16 x (mfma + 2 v_cvt_pk)    [alone 16+8.8]", 48, w, dout, sink);
    run<24>("16 x (mfma + 2 v_max3_f32)  [alone 16+8.6]", 48, w, dout, sink);
    run<30>("16 x ds_read_b128 + wait", 16, w, dout, sink);
    run<31>("16 x ds_read_b64_tr_b16 + wait", 16, w, dout, sink);
    run<32>("16 x (mfma + 1 ds_read_b128)", 32, w, dout, sink);
    run<33>("16 x (mfma + 2 ds_read_b128)", 48, w, dout, sink);
    run<34>("16 x (mfma + 2 ds_read_b64_tr_b16)", 48, w, dout, sink);
    run<35>("16 x mfma + 1 s_barrier", 17, w, dout, sink);
    run<36>("4 x (4 mfma + s_barrier)", 20, w, dout, sink);
    run<18>("attention tile, plain instead of packed: vector part alone (160)", 160, w, dout, sink);
    run<17>("attention tile, plain instead of packed: 16 mfma | vector | 16 mfma", 192, w, dout, sink);
    run<16>("attention tile: vector part alone (120)", 120, w, dout, sink);
    run<14>("attention tile: 16 mfma | vector | 16 mfma", 152, w, dout, sink);
    run<15>("attention tile: interleaved", 152, w, dout, sink);
  }
  return 0;
}
