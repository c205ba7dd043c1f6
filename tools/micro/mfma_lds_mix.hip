// How much of the persistent GEMM's K loop is its instruction MIX? The loop of gemm256p_f16_kernel issues, per wave and 64-deep K step, 64 MFMAs
// (16x16x32 f16), 24 ds_read_b128 fragment reads, 8 LDS-DMA requests of 1 KiB and one workgroup barrier. This program runs that mix with no
// data dependence on real tiles (operands come from LDS reads issued half a step earlier, like the kernel; the DMA refills an LDS ring from an
// L2-resident buffer) and reports matrix-pipe utilisation for: MFMAs alone / + fragment reads / + DMA / + the barrier.
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_mix mfma_lds_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

template <bool READS, bool DMA, bool BARRIER>
__global__ __launch_bounds__(512) void mix(const half8* __restrict__ g, float* out, unsigned long long* clk, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half8* lds = reinterpret_cast<half8*>(smem);   // 2 slots x 64 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = g[i & 4095];
  __syncthreads();
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  half8 w0[4], x0[8], w1[4], x1[8];
  for (int t = 0; t < 4; ++t) w0[t] = w1[t] = lds[(lane + 64 * t) & 4095];
  for (int t = 0; t < 8; ++t) x0[t] = x1[t] = lds[(lane + 64 * (t + 4)) & 4095];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < steps; ++s) {
    const half8* cur = lds + (s & 1) * 4096;
    const half8* nxt = lds + ((s + 1) & 1) * 4096;
    // half 0: MFMAs on (w0, x0); fragment reads of half 1, two per group of 4 MFMAs
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[gq][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], x0[gq], acc[gq][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (READS) {
        if (gq < 2) {
          w1[2 * gq] = cur[(lane + 64 * (2 * gq) + 1024) & 4095];
          w1[2 * gq + 1] = cur[(lane + 64 * (2 * gq + 1) + 1024) & 4095];
        } else if (gq < 6) {
          x1[2 * (gq - 2)] = cur[(lane + 64 * (2 * (gq - 2)) + 2048) & 4095];
          x1[2 * (gq - 2) + 1] = cur[(lane + 64 * (2 * (gq - 2) + 1) + 2048) & 4095];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (BARRIER) __builtin_amdgcn_s_barrier();
    // half 1: MFMAs on (w1, x1); one DMA request and the reads of the next step's half 0 per group
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[gq][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[nt], x1[gq], acc[gq][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (DMA)
        __builtin_amdgcn_global_load_lds((const GLB_AS void*)(g + ((s * 64 + gq * 8 + wave) & 63) * 64 + lane),
                                         (LDS_AS void*)(const_cast<half8*>(cur) + (gq * 8 + wave) * 64), 16, 0, 0);
      if (READS) {
        if (gq < 2) {
          w0[2 * gq] = nxt[(lane + 64 * (2 * gq)) & 4095];
          w0[2 * gq + 1] = nxt[(lane + 64 * (2 * gq + 1)) & 4095];
        } else if (gq < 6) {
          x0[2 * (gq - 2)] = nxt[(lane + 64 * (2 * (gq - 2)) + 512) & 4095];
          x0[2 * (gq - 2) + 1] = nxt[(lane + 64 * (2 * (gq - 2) + 1) + 512) & 4095];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
  if (sum == 12345.678f) out[blockIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

template <bool READS, bool DMA, bool BARRIER>
static void run(const char* label, const half8* g, float* out, unsigned long long* clk, int steps) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(mix<READS, DMA, BARRIER>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((mix<READS, DMA, BARRIER>), dim3(256), dim3(512), 131072, 0, g, out, clk, steps);
  hipEventRecord(e0);
  hipLaunchKernelGGL((mix<READS, DMA, BARRIER>), dim3(256), dim3(512), 131072, 0, g, out, clk, steps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);
  const double cyc = ms * 1e6 * ghz / steps;   // cycles per K step and SIMD (two waves, 64 MFMAs each = 2048 pipe cycles)
  printf("%-52s %7.0f cycles per K step and SIMD (pipe 2048: %.0f %% busy), clock %.2f GHz, %.3f ms, %.0f TFLOP/s\n", label, cyc, 204800.0 / cyc, ghz, ms,
         256.0 * 8 * steps * 64 * 16384.0 / ms / 1e9);
}

int main() {
  half8* g;
  float* out;
  unsigned long long* clk;
  hipMalloc(&g, 4096 * sizeof(half8));
  hipMalloc(&out, 4096);
  hipMalloc(&clk, 64);
  _Float16 h[4096 * 8];
  for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)(0.37f * (float)((i * 7919) % 13 - 6));
  (void)hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  const int steps = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    run<false, false, false>("MFMAs only", g, out, clk, steps);
    run<true, false, false>("+ 24 ds_read_b128 per wave-step", g, out, clk, steps);
    run<true, true, false>("+ 8 LDS-DMA requests per wave-step (vmcnt(0) mid-step)", g, out, clk, steps);
    run<true, true, true>("+ one workgroup barrier per step (the kernel's mix)", g, out, clk, steps);
    run<true, false, true>("reads + barrier, no DMA", g, out, clk, steps);
  }
  return 0;
}
