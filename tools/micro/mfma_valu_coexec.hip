// Can vector-ALU instructions run UNDER f16 MFMAs on gfx950 -- inside one wave, and between the two waves that share a SIMD -- and does it
// depend on the MFMA shape? (Round 4: every restructuring of the three-pass attention that put VALU work beside MFMAs measured nothing; its
// PMC counters show matrix-pipe time and vector-issue time ADDING UP.) Register operands only, no LDS / global traffic.
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coexec mfma_valu_coexec.hip       run: ./mfma_valu_coexec
// Per loop iteration a wave issues NM MFMAs (round-robin over independent accumulators: no dependent back-to-back pair) and NV plain
// v_fma_f32 (8 independent chains), evenly interleaved, pinned with sched_barrier. Modes:
//   M only / V only / M+V in ONE wave per SIMD / M+V in BOTH waves of a SIMD / wave A = M only beside wave B = V only on the same SIMD.
// Output: cycles per iteration (wall time x the clock measured by s_memtime / s_memrealtime inside the kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ROLE: 0 = every wave runs the interleaved stream; 1 = waves 0-3 MFMAs only, waves 4-7 VALU only (8-wave workgroups: w and w + 4 share a SIMD)
template <int SHAPE, int NM, int NV, int ROLE>
__global__ __launch_bounds__(512) void coexec(float* out, unsigned long long* clk, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  half8 a[2], b[2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (_Float16)(seed * (float)((threadIdx.x * 7 + i * 3 + j) % 13 - 6));
      b[i][j] = (_Float16)(seed * (float)((threadIdx.x * 5 + i * 11 + j) % 17 - 8));
    }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed * (float)(threadIdx.x % 7 + i);
  const float c0 = 1.0f + seed * 1e-3f, c1 = seed * 1e-4f;
  f32x4 acc4[4];
  f32x16 acc16[4];
  for (int i = 0; i < 4; ++i) {
    acc4[i] = f32x4{0, 0, 0, 0};
    for (int r = 0; r < 16; ++r) acc16[i][r] = 0.f;
  }
  const bool do_m = ROLE == 0 || wave < 4, do_v = ROLE == 0 || wave >= 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    constexpr int G = NM > 0 ? NM : 1;  // groups: one MFMA followed by NV / NM fmas
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (NM > 0 && do_m) {
        if (SHAPE == 16) acc4[g & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[g & 1], b[(g >> 1) & 1], acc4[g & 3], 0, 0, 0);
        else acc16[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[g & 1], b[(g >> 1) & 1], acc16[g & 3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (NV > 0 && do_v) {
#pragma unroll
        for (int k = 0; k < NV / G; ++k) {
          const int r = (g * (NV / G) + k) & 7;
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(c0), "v"(c1));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc16[i][0] + acc16[i][15];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) out[blockIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = t1 - t0;   // core clock ticks
    clk[1] = r1 - r0;   // 100 MHz ticks
  }
}

template <int SHAPE, int NM, int NV, int ROLE>
static void run(const char* label, int block, float* out, unsigned long long* clk, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((coexec<SHAPE, NM, NV, ROLE>), dim3(256), dim3(block), 0, 0, out, clk, iters, 0.37f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((coexec<SHAPE, NM, NV, ROLE>), dim3(256), dim3(block), 0, 0, out, clk, iters, 0.37f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);   // core ticks per ns
  const double cyc = (double)h[0] / iters;
  printf("%-44s %dx%d  %2d MFMA + %2d v_fma per iteration, %d waves/SIMD: %7.1f cycles / iteration (in-kernel clock %.2f GHz, %.3f ms)\n", label, SHAPE, SHAPE, NM, NV,
         block / 256, cyc, ghz, ms);
}

int main() {
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, 4096);
  hipMalloc(&clk, 64);
  const int it = 20000;
  // 16x16x32: 16 pipe cycles each; 32x32x16: 32. Equal pipe time per iteration: 8 x 16 = 4 x 32 = 128 cycles; 24 v_fma = 96 issue cycles (4 each)
  run<16, 8, 0, 0>("MFMA only", 256, out, clk, it);
  run<16, 0, 24, 0>("VALU only", 256, out, clk, it);
  run<16, 8, 24, 0>("interleaved in one wave", 256, out, clk, it);
  run<16, 8, 24, 0>("interleaved, both waves of a SIMD", 512, out, clk, it);
  run<16, 8, 24, 1>("wave A MFMA only beside wave B VALU only", 512, out, clk, it);
  run<16, 8, 0, 0>("MFMA only, both waves", 512, out, clk, it);
  run<16, 0, 24, 0>("VALU only, both waves", 512, out, clk, it);
  run<32, 4, 0, 0>("MFMA only", 256, out, clk, it);
  run<32, 4, 24, 0>("interleaved in one wave", 256, out, clk, it);
  run<32, 4, 24, 0>("interleaved, both waves of a SIMD", 512, out, clk, it);
  run<32, 4, 24, 1>("wave A MFMA only beside wave B VALU only", 512, out, clk, it);
  run<32, 4, 0, 0>("MFMA only, both waves", 512, out, clk, it);
  // denser vector work: 48 v_fma per 128 pipe cycles
  run<16, 8, 48, 0>("interleaved in one wave", 256, out, clk, it);
  run<32, 4, 48, 0>("interleaved in one wave", 256, out, clk, it);
  run<16, 8, 48, 0>("interleaved, both waves of a SIMD", 512, out, clk, it);
  run<32, 4, 48, 0>("interleaved, both waves of a SIMD", 512, out, clk, it);
  return 0;
}
