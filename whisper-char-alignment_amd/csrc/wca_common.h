// Shared device/host helpers for the MI355X (gfx950) forced-alignment engine.
// Everything here is written for CDNA4 only: 64-wide wavefronts, MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wca {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

#define WCA_LDS __attribute__((address_space(3)))
#define WCA_GLOBAL __attribute__((address_space(1)))

constexpr int kWave = 64;

// ---- LDS-DMA: 16 bytes per lane, LDS destination = wave-uniform base + lane*16 ----
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const WCA_GLOBAL void*)gsrc, (WCA_LDS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Swizzle for [rows][64 x f16] (128-byte rows) LDS tiles read with ds_read_b128:
// 16-byte chunk c of row r lives at chunk position c ^ swz(r). Conflict-free for the
// MFMA operand reads used in gemm.hip / attention.hip (checked against the gfx950
// ds_read_b128 lane groups).
__device__ __forceinline__ int swz128(int r) { return ((r >> 1) ^ (r >> 3)) & 7; }

// Wave-wide (64-lane) all-reduce without LDS traffic: four DPP row rotations reduce each 16-lane row, then
// v_permlane16_swap / v_permlane32_swap (which hand back {own, partner} in some order) combine the four rows.
// Requires all 64 lanes active.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0x121>(v);  // row_ror:1
  v += dpp_mov<0x122>(v);  // row_ror:2
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x128>(v);  // row_ror:8
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0x121>(v));
  v = fmaxf(v, dpp_mov<0x122>(v));
  v = fmaxf(v, dpp_mov<0x124>(v));
  v = fmaxf(v, dpp_mov<0x128>(v));
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// Exact-erf GELU, GELU(x) = x * Phi(x), evaluated as  max(x, 0) - | 0.5 * x * erfc(|x| / sqrt(2)) |  with erfc from
// Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 on erf): t = 1 / (1 + p|x|/sqrt2), erfc = poly(t) * exp(-x^2/2). The 0.5
// and the sqrt(2) are folded into the constants; one v_rcp + one v_exp + 9 plain VALU ops per element. Measured
// against float64 erf over [-12, 12]: max abs error 3.3e-7, max relative error 1.7e-4 (below the 2^-11 rounding
// of the f16 store that follows), i.e. this is the erf GELU of whisper.model, not the tanh approximation.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.2316418880f, 1.0f));
  float p = fmaf(0.5307027145f, t, -0.7265760135f);
  p = fmaf(p, t, 0.7107068705f);
  p = fmaf(p, t, -0.1422483680f);
  p = fmaf(p, t, 0.1274147960f);
  const float e = __builtin_amdgcn_exp2f(x * x * -0.7213475204f);
  const float h = p * t * e * x;
  return fmaxf(x, 0.f) - fabsf(h);
}

// Reference-precision ("split") mode: torch's erf GELU, x * 0.5 * (1 + erf(x / sqrt 2)), to fp32 accuracy -- the polynomial above is exact
// to the f16 store that follows it in the default mode; in the pair-output epilogues the value is kept to ~22 bits, so its 3e-7 absolute
// error would show. Until round 4 this was the device library's erff (<= 2 ulp, but ~45 vector instructions per element once both of its
// branches are if-converted: the GELU epilogue was 15-17 % of the pair fc1 launch); now, on TWO elements with packed fp32 instructions:
//   GELU(x) = max(x, 0) - |x| h,   h = erfc(|x| / sqrt 2) / 2 = t exp(-z^2 + P(t)) / 2,   z = |x| / sqrt 2,  t = 1 / (1 + z / 2)
// with P the degree-9 Chebyshev fit of Numerical Recipes' erfcc (fractional error of erfc < 1.2e-7 EVERYWHERE, so the negative tail keeps
// its relative accuracy -- the reference's own fp32 expression x 0.5 (1 + erf(x / sqrt 2)) loses it to cancellation). log2(e) and the
// factor 1/2 are folded into the coefficients (exp2 of the scaled argument - 1). ~13.5 vector issue slots per element: |x|, 1 + z/2, v_rcp,
// nine packed fmas, x^2, the exponent, v_exp, t e, max(x, 0), |x| h, the difference. Against float64 over [-12, 12] in emulated fp32: absolute
// error <= 1.1e-7 on |x| < 1, <= 2.1e-7 on 1 <= |x| < 3 (the fp32 rounding of the result itself: the reference formula measures 0.8e-7 / 2.5e-7),
// relative error <= 4.7e-7 on [-1, 1], <= 1.4e-6 down to -3.
__device__ __forceinline__ f32x2 gelu_erfc2(f32x2 x) {
  constexpr float L2E = 1.4426950408889634f;
  const f32x2 ax = f32x2{__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
  const f32x2 d = __builtin_elementwise_fma(ax, f32x2{0.35355339059327373f, 0.35355339059327373f}, f32x2{1.0f, 1.0f});   // 1 + |x| / (2 sqrt 2)
  const f32x2 t = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2 p = f32x2{0.17087277f * L2E, 0.17087277f * L2E};
#define WCA_GH(C) p = __builtin_elementwise_fma(p, t, f32x2{(C) * L2E, (C) * L2E})
  WCA_GH(-0.82215223f);
  WCA_GH(1.48851587f);
  WCA_GH(-1.13520398f);
  WCA_GH(0.27886807f);
  WCA_GH(-0.18628806f);
  WCA_GH(0.09678418f);
  WCA_GH(0.37409196f);
  WCA_GH(1.00002368f);
#undef WCA_GH
  p = __builtin_elementwise_fma(p, t, f32x2{-1.26551223f * L2E - 1.0f, -1.26551223f * L2E - 1.0f});   // ... and the factor 1/2
  const f32x2 w = x * x;
  const f32x2 arg = __builtin_elementwise_fma(w, f32x2{-0.5f * L2E, -0.5f * L2E}, p);                     // log2(e) (P(t) - z^2) - 1
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
  const f32x2 h = t * e;
  const f32x2 pos = f32x2{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
  return pos - ax * h;
}

// x -> (hi, lo) with hi = f16(x), lo = f16(x - hi): x = hi + lo up to 2^-22 |x| (lo is exact in f32; it rounds to f16 with
// 11 more bits, or to the f16 subnormal grid, 6e-8 absolute, for |x| below ~0.1). The value is PINNED in a register first:
// with -ffp-contract=fast (the HIP default) hipcc otherwise folds the multiply that produced x into ONE of the two conversions
// (v_fma_mixlo_f16 rounds the exact product once, the other use rounds the fp32 value), and at an exact f16 tie the stored hi and
// the hi under the subtraction are different neighbours -- lo then has the wrong sign (measured: one element in ~8000, error 1 ulp
// of hi).
struct HalfPair {
  half_t hi, lo;
};
__device__ __forceinline__ HalfPair split_pair(float x) {
  asm volatile("" : "+v"(x));
  HalfPair r;
  r.hi = (half_t)x;
  r.lo = (half_t)(x - (float)r.hi);
  return r;
}

// Two values at once: hi = f16(x) by the packed RNE conversion (v_cvt_pk_f16_f32), lo = f16(x - hi) by the mixed-precision fma
// (v_fma_mixlo / mixhi_f16: f32 x, f16 hi operand picked by op_sel; x - hi is exact before its one rounding) -- 3 vector instructions for two
// elements where split_pair() costs 8 and a pack. Same values as split_pair(): the inputs are pinned first for the same reason.
struct Half2Pair {
  half2_ hi, lo;
};
__device__ __forceinline__ Half2Pair split_pair2(float x0, float x1) {
  asm volatile("" : "+v"(x0), "+v"(x1));
  Half2Pair r;
  r.hi = __builtin_convertvector((f32x2{x0, x1}), half2_);
  const unsigned hu = __builtin_bit_cast(unsigned, r.hi);
  unsigned lu;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lu) : "v"(x0), "v"(hu));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lu) : "v"(x1), "v"(hu));
  r.lo = __builtin_bit_cast(half2_, lu);
  return r;
}

// Bijective XCD-aware remap of a linear workgroup id: blocks that share an XCD
// (id % 8 equal) get a contiguous range of logical ids, so neighbouring tiles hit one L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = orig & 7, idx = orig >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace wca
