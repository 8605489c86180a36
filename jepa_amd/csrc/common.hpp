// Shared device/host helpers for the V-JEPA gfx950 kernels.
// Everything here is CDNA4-only (wave64, MFMA, LDS-DMA); there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits in HBM

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define VJ_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved: identical to torch's float -> bfloat16
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// two fp32 -> packed bf16x2 (RNE); lowers to one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  bf2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// the same, opaque to the optimiser: when only the UNPACKED halves of the result are used, hipcc otherwise converts each
// element on its own (v_cvt_pk_bf16_f32 with a dummy partner) -- twice the conversions
__device__ __forceinline__ uint32_t pack_bf2_opaque(float lo, float hi) {
  uint32_t w = pack_bf2(lo, hi);
  asm("" : "+v"(w));
  return w;
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15); every lane of the row receives the total.  Rotate-and-add with
// row_ror: four v_add_f32 with a DPP source, no LDS round trip (ds_bpermute: ~24 cycles of the LDS pipe each).
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf-GELU (nn.GELU() default, reference modules.py:32) and its derivative.  erf by Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7, far below the bf16 rounding of the result): one v_rcp + one v_exp + 6 FMAs instead of
// libm's ~40-instruction erff -- the fc1 epilogue applies this to every element of the 4D-wide hidden layer.
// The exponential exp(-x^2/2) is shared between erf(x/sqrt2) and the Gaussian term of the derivative.
__device__ __forceinline__ void erf_core(float x, float& erf_v, float& gauss) {
  const float z = fabsf(x) * 0.70710678118654752f;          // |x|/sqrt(2)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  gauss = __expf(-z * z);                                     // exp(-x^2/2)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * gauss;
  erf_v = copysignf(e, x);
}
__device__ __forceinline__ float gelu_f(float x) {
  float e, g;
  erf_core(x, e, g);
  return 0.5f * x * (1.0f + e);
}
__device__ __forceinline__ float dgelu_f(float x) {
  float e, g;
  erf_core(x, e, g);
  return 0.5f * (1.0f + e) + x * 0.3989422804014327f * g;
}

// ---- 2-wide versions for the GEMM epilogues: the polynomial, the products and the final combination issue as
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two elements per instruction); only rcp / exp2 / the sign handling stay
// scalar.  q = 0.5 * erfc(|x|/sqrt2) = 0.5 * poly(t) * t * exp(-x^2/2) (same A-S 7.1.26 coefficients, halved).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void half_erfc2(f32x2_t x, f32x2_t& q, f32x2_t& gauss) {
  const float k = 0.3275911f * 0.70710678118654752f;
  f32x2_t t = {__builtin_amdgcn_rcpf(fmaf(k, fabsf(x[0]), 1.0f)), __builtin_amdgcn_rcpf(fmaf(k, fabsf(x[1]), 1.0f))};
  const f32x2_t c = {-0.72134752044448170f, -0.72134752044448170f};   // -0.5 * log2(e)
  const f32x2_t a = (x * c) * x;
  gauss = (f32x2_t){__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};   // exp(-x^2/2)
  const f32x2_t c5 = {0.5f * 1.061405429f, 0.5f * 1.061405429f}, c4 = {0.5f * -1.453152027f, 0.5f * -1.453152027f},
                 c3 = {0.5f * 1.421413741f, 0.5f * 1.421413741f}, c2 = {0.5f * -0.284496736f, 0.5f * -0.284496736f},
                 c1 = {0.5f * 0.254829592f, 0.5f * 0.254829592f};
  f32x2_t p = __builtin_elementwise_fma(c5, t, c4);
  p = __builtin_elementwise_fma(p, t, c3);
  p = __builtin_elementwise_fma(p, t, c2);
  p = __builtin_elementwise_fma(p, t, c1);
  q = (p * t) * gauss;
}
// relu on the bit pattern: negative floats are negative integers (one v_max_i32; fmaxf() on a value that was assembled
// from bits costs a second, canonicalising v_max_f32).  NaN passes through or becomes 0 -- the callers' inputs are finite.
__device__ __forceinline__ float relu_bits(float x) {
  const int b = __float_as_int(x);
  return __int_as_float(b > 0 ? b : 0);
}
// gelu(x) = x * Phi(x) = max(x, 0) - |x * q|   (q = 0.5 erfc(|x|/sqrt2) >= 0; relative accuracy is kept in the negative tail)
// (a single v_fma_f32 with -|x| as a source modifier would save half an instruction per element, but hipcc packs the two
//  FMAs into a v_pk_fma_f32, which has no |x| modifier, and forcing the scalar form through inline assembly makes the
//  one-pass epilogue of gemm8.hip spill 140 VGPRs)
__device__ __forceinline__ f32x2_t gelu2(f32x2_t x) {
  f32x2_t q, g;
  half_erfc2(x, q, g);
  const f32x2_t h = x * q;
  return (f32x2_t){relu_bits(x[0]) - fabsf(h[0]), relu_bits(x[1]) - fabsf(h[1])};
}
// gelu(x) AND gelu'(x) from one evaluation of the erfc core: the fc1 epilogue of a layer that will run backward stores
// gelu'(u) (bf16) instead of the pre-activation u, and the fc2 dgrad epilogue only multiplies by it -- the backward never
// re-evaluates exp / rcp / the polynomial (u has no other consumer)
__device__ __forceinline__ void gelu_dgelu2(f32x2_t x, f32x2_t& y, f32x2_t& d) {
  f32x2_t q, g;
  half_erfc2(x, q, g);
  const f32x2_t h = x * q;
  y = (f32x2_t){relu_bits(x[0]) - fabsf(h[0]), relu_bits(x[1]) - fabsf(h[1])};   // = gelu2(x), bit for bit
  const f32x2_t half = {0.5f, 0.5f}, a = half - q;
  const f32x2_t phi = half + (f32x2_t){copysignf(a[0], x[0]), copysignf(a[1], x[1])};
  const f32x2_t c = {0.3989422804014327f, 0.3989422804014327f};
  d = __builtin_elementwise_fma(x * c, g, phi);                                    // = dgelu2(x), bit for bit
}
// ---- round 4: q = Phi(-|x|) without the reciprocal (option gelu_poly, default) --------------------------------------------
// log2 Phi(-a) is a smooth, nearly quadratic function of a >= 0 (-1 at 0, ~ -a^2 log2(e)/2 - log2(a sqrt(2 pi)) far out), so
// q(a) = exp2(L(a)) with a degree-6 minimax polynomial L on [0, 5] (Lawson iteration, max |dL| 1.8e-5, i.e. q to 1.3e-5
// RELATIVE over the whole range, tail included) replaces A-S 7.1.26's v_rcp + 4 FMAs + 3 multiplies by 6 FMAs; the one v_exp
// stays.  a = min(|x|, 5) is ONE instruction (v_min_f32 with the |x| source modifier) and is used for the product a * q as
// well: beyond 5 the result is relu(x) - 5 q(5) = relu(x) - 1.4e-6 (exact: relu(x) - |x| Phi(-|x|), at most 1.4e-6 there).
// Against the correctly rounded bf16 erf-GELU over ALL finite bf16 inputs x > -5: 5 of 20712 results differ (by one bf16 ulp);
// the A-S form: 22 (tests/test_gelu_poly.py enumerates them on the CPU with this arithmetic).  relu(x) is taken as
// 0.5 * (x + |x|): exact for finite x, and a NaN of EITHER sign propagates (relu_bits drops a negative-signed NaN, which
// the A-S form only survives because its reciprocal carries the NaN into q).
// x + |x| = 2 relu(x) as ONE v_add_f32 with the |x| source modifier (hipcc turns the C expression into v_and + a packed add);
// exact for |x| < 2^127 (no cancellation error for negative x, unlike 0.5 x + 0.5 |x| - ...), +inf above; a NaN of either sign stays a NaN
__device__ __forceinline__ float twice_relu(float x) {
  float s;
  asm("v_add_f32 %0, %1, |%1|" : "=v"(s) : "v"(x));
  return s;
}
__device__ __forceinline__ f32x2_t half_erfc2_lp(f32x2_t x, f32x2_t& a) {
  // v_med3_f32 a, |x|, 0, 5: one instruction (fminf(fabsf(x), 5) costs a canonicalising v_max + v_min + v_and)
  a = (f32x2_t){__builtin_amdgcn_fmed3f(__builtin_fabsf(x[0]), 0.0f, 5.0f), __builtin_amdgcn_fmed3f(__builtin_fabsf(x[1]), 0.0f, 5.0f)};
  const f32x2_t c6 = {2.945814386e-05f, 2.945814386e-05f}, c5 = {-7.087827263e-04f, -7.087827263e-04f},
                 c4 = {7.746013931e-03f, 7.746013931e-03f}, c3 = {-5.260629358e-02f, -5.260629358e-02f},
                 c2 = {-4.596254594e-01f, -4.596254594e-01f}, c1 = {-1.150867238e+00f, -1.150867238e+00f},
                 c0 = {-1.000017643e+00f, -1.000017643e+00f};
  f32x2_t p = __builtin_elementwise_fma(c6, a, c5);
  p = __builtin_elementwise_fma(p, a, c4);
  p = __builtin_elementwise_fma(p, a, c3);
  p = __builtin_elementwise_fma(p, a, c2);
  p = __builtin_elementwise_fma(p, a, c1);
  p = __builtin_elementwise_fma(p, a, c0);
  return (f32x2_t){__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
}
// gelu(x) = relu(x) - a q(a)
__device__ __forceinline__ f32x2_t gelu2_lp(f32x2_t x) {
  f32x2_t a;
  const f32x2_t q = half_erfc2_lp(x, a);
  const f32x2_t s = {twice_relu(x[0]), twice_relu(x[1])};
  const f32x2_t half = {0.5f, 0.5f};
  return __builtin_elementwise_fma(s, half, -(a * q));
}
// gelu(x) and gelu'(x) = Phi(x) + x pdf(x): Phi from the same q, the Gaussian term from its own v_exp
__device__ __forceinline__ void gelu_dgelu2_lp(f32x2_t x, f32x2_t& y, f32x2_t& d) {
  f32x2_t a;
  const f32x2_t q = half_erfc2_lp(x, a);
  const f32x2_t s = {twice_relu(x[0]), twice_relu(x[1])};
  const f32x2_t half = {0.5f, 0.5f};
  y = __builtin_elementwise_fma(s, half, -(a * q));                                  // = gelu2_lp(x), bit for bit
  const f32x2_t c = {-0.72134752044448170f, -0.72134752044448170f};                  // -0.5 * log2(e)
  const f32x2_t e = (x * c) * x;
  const f32x2_t g = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};    // exp(-x^2/2)
  const f32x2_t hq = half - q;
  const f32x2_t phi = half + (f32x2_t){copysignf(hq[0], x[0]), copysignf(hq[1], x[1])};
  const f32x2_t k = {0.3989422804014327f, 0.3989422804014327f};
  d = __builtin_elementwise_fma(x * k, g, phi);
}
// gelu'(x) = Phi(x) + x * pdf(x),  Phi(x) = 0.5 + copysign(0.5 - q, x)
__device__ __forceinline__ f32x2_t dgelu2(f32x2_t x) {
  f32x2_t q, g;
  half_erfc2(x, q, g);
  const f32x2_t half = {0.5f, 0.5f}, a = half - q;
  const f32x2_t phi = half + (f32x2_t){copysignf(a[0], x[0]), copysignf(a[1], x[1])};
  const f32x2_t c = {0.3989422804014327f, 0.3989422804014327f};
  return __builtin_elementwise_fma(x * c, g, phi);
}

// ---- per-device one-time setup ------------------------------------------------------------------------------------
// Function attributes (dynamic LDS limit), event pools and small constant buffers belong to a DEVICE, not to the process:
// a host that drives several GPUs from one process (not this package's own launcher: one process per GPU) calls the C ABI
// with different current devices.  vj_device_slot() = current HIP device (0 on error); VjPerDeviceOnce runs an idempotent
// setup once per device (a lost race repeats it harmlessly).
#include <atomic>
#define VJ_MAX_DEVICES 64
inline int vj_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev % VJ_MAX_DEVICES;
}
struct VjPerDeviceOnce {
  std::atomic<unsigned long long> done{0};
  template <class F>
  void operator()(F&& f) {
    const unsigned long long bit = 1ull << vj_device_slot();
    if (!(done.load(std::memory_order_acquire) & bit)) {
      f();
      done.fetch_or(bit, std::memory_order_release);
    }
  }
};

// ---- host side error plumbing (no C++ exception crosses the C ABI) ----
extern "C" const char* vj_last_error(void);
void vj_set_error(const char* fmt, ...);

#define VJ_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      vj_set_error(__VA_ARGS__);           \
      return -1;                           \
    }                                      \
  } while (0)

#define VJ_LAUNCH_CHECK(name)                                              \
  do {                                                                     \
    hipError_t _e = hipGetLastError();                                     \
    if (_e != hipSuccess) {                                                \
      vj_set_error("%s: launch failed: %s", name, hipGetErrorString(_e)); \
      return (int)_e;                                                      \
    }                                                                      \
  } while (0)

__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
