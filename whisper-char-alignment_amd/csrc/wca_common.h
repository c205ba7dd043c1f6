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

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. below f32 epsilon of the 0.5*(1+erf) factor):
// the GELU result is rounded to f16 (2^-11) right after, so this is "exact (erf) GELU" to the last stored bit
// in all but measure-zero rounding ties, at a third of the instructions of libm's erff.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float r = 1.0f - p * t * e;
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
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
