// Sustained f16 MFMA rate under the MI355X power cap: v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16, register operands only
// (no LDS / global traffic), 2 waves per SIMD, ~1 ms of back-to-back MFMAs per launch. build: hipcc --offload-arch=gfx950 -O3 -o mfma_shapes mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, float seed) {
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (_Float16)(seed * (float)((threadIdx.x * 7 + i * 3 + j) % 13 - 6));
      b[i][j] = (_Float16)(seed * (float)((threadIdx.x * 5 + i * 11 + j) % 17 - 8));
    }
  float s = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  }
  if (s == 12345.678f) out[blockIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256;  // one 8-wave workgroup per CU: 2 waves per SIMD
  for (int rep = 0; rep < 3; ++rep)
    for (int shape : {16, 32}) {
      for (float seed : {0.0f, 0.37f}) {
        const int iters = shape == 16 ? 8000 : 8000;  // 16 x 16x16x32 (16 cyc) = 8 x 32x32x16 (32 cyc) = 256 pipe cycles per iteration
        for (int warm = 0; warm < 2; ++warm) {
          if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(grid), dim3(512), 0, 0, out, iters, seed);
          else hipLaunchKernelGGL(mfma_loop<32>, dim3(grid), dim3(512), 0, 0, out, iters, seed);
        }
        hipEventRecord(e0);
        for (int k = 0; k < 5; ++k) {
          if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(grid), dim3(512), 0, 0, out, iters, seed);
          else hipLaunchKernelGGL(mfma_loop<32>, dim3(grid), dim3(512), 0, 0, out, iters, seed);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        const double flops = (double)grid * 8 * iters * (shape == 16 ? 16 * 16384.0 : 8 * 32768.0);
        printf("shape %dx%d %s operands: %.3f ms per launch, %.0f TFLOP/s (%.1f %% of 2.5 PF)\n", shape, shape, seed == 0.f ? "zero  " : "random", ms, flops / ms / 1e9,
               flops / ms / 1e9 / 25.0);
      }
    }
  return 0;
}
