// DTW cost-table fill + backtrace on gfx950.  Reference call site: timing.py:103 `dtw(-matrix)`, which
// from force_align always reaches openai-whisper's numba `dtw_cpu` (matrix was moved to the CPU at
// timing.py:102) followed by `backtrace`.  This kernel reproduces THAT arithmetic bit for bit:
//   c0 = C[i-1][j-1], c1 = C[i-1][j], c2 = C[i][j-1]
//   diag only if strictly smallest, else up only if strictly smallest, else left (every tie -> left)
//   C[i][j] = float( double(x[i][j]) + double(c) )        (f32 table fed by f64 x, x = -matrix)
//
// Mapping: ONE 64-lane wave per problem, no LDS, no barriers.  Lane l owns R = ceil(N/64) consecutive
// text rows and sweeps them along the frame axis, skewed by one column per lane (lane l is at column
// s - l in step s): a wavefront-parallel anti-diagonal sweep whose only cross-lane traffic is the
// previous lane's last row (two values per step).  The 2-bit move trace is packed 16 columns per word.
// It is neither HBM- nor MFMA-bound: N+M-1 dependent steps, reported as microseconds per utterance.
#include "kernels.h"
#include "wca_common.h"

namespace wca {

namespace {

template <int R, int U>
__global__ __launch_bounds__(64) void dtw_kernel(DtwArgs a) {
  const int p = blockIdx.x;
  const int lane = threadIdx.x;
  const int N = a.N ? a.N[p] : a.N_all;
  const int M = a.M ? a.M[p] : a.M_all;
  const int cap = a.N_max + a.M_max + 2;
  int* pt = a.path + (long)p * 2 * cap;
  int* pj = pt + cap;
  if (N <= 0 || M <= 0 || N > 64 * R || N > a.N_max || M > a.M_max) {
    if (lane == 0) a.path_len[p] = 0;
    return;
  }
  const float* __restrict__ x = a.matrix + (long)p * a.m_bs;
  const int wpr = (a.M_max + 15) >> 4;
  uint32_t* __restrict__ tr = a.trace + (long)p * a.N_max * wpr;

  const int i0 = lane * R;
  float prev[R];
  uint32_t tw[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    prev[r] = INFINITY;
    tw[r] = 0u;
  }
  float last_cur = INFINITY, last_prev = INFINITY;
  const int lane_max = (N - 1) / R;
  const int S = M + lane_max;  // steps 0 .. S-1

  for (int s0 = 0; s0 < S; s0 += U) {
    // loads of the whole block first (addresses do not depend on the recurrence)
    float xv[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int j = s0 + u - lane;
      j = j < 0 ? 0 : (j >= M ? M - 1 : j);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        int i = i0 + r;
        i = i < N ? i : N - 1;
        xv[u][r] = x[(long)i * a.ld + j];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = s0 + u;
      const int j = s - lane;
      float up_in = __shfl_up(last_cur, 1);
      float diag_in = __shfl_up(last_prev, 1);
      if (lane == 0) {
        up_in = INFINITY;
        diag_in = (j == 0) ? 0.f : INFINITY;
      }
      const bool active = (s < S) && (j >= 0) && (j < M) && (i0 < N);
      if (active) {
        float newv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int i = i0 + r;
          const float c1 = (r == 0) ? up_in : newv[r > 0 ? r - 1 : 0];
          const float c0 = (r == 0) ? diag_in : prev[r > 0 ? r - 1 : 0];
          const float c2 = prev[r];
          float c;
          uint32_t t;
          if (c0 < c1 && c0 < c2) {
            c = c0;
            t = 0u;
          } else if (c1 < c0 && c1 < c2) {
            c = c1;
            t = 1u;
          } else {
            c = c2;
            t = 2u;
          }
          const float nv = (float)((double)(-xv[u][r]) + (double)c);
          newv[r] = (i < N) ? nv : INFINITY;
          if (i < N) {
            tw[r] |= t << (2 * (j & 15));
            if ((j & 15) == 15 || j == M - 1) {
              tr[(long)i * wpr + (j >> 4)] = tw[r];
              tw[r] = 0u;
            }
          }
        }
        last_prev = prev[R - 1];
        last_cur = newv[R - 1];
#pragma unroll
        for (int r = 0; r < R; ++r) prev[r] = newv[r];
      }
    }
  }

  __threadfence();  // trace words written by all lanes -> visible to lane 0's loads below
  if (lane == 0) {
    int i = N - 1, j = M - 1, k = cap;
    int* jf = a.jump_frame ? a.jump_frame + (long)p * a.jump_ld : nullptr;
    while ((i >= 0 || j >= 0) && k > 0) {
      --k;
      pt[k] = i;
      pj[k] = j;
      if (jf && i >= 0) jf[i] = j;
      int t;
      if (i < 0) {
        t = 2;
      } else if (j < 0) {
        t = 1;
      } else {
        const uint32_t w = __hip_atomic_load(tr + (long)i * wpr + (j >> 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (int)((w >> (2 * (j & 15))) & 3u);
      }
      if (t == 0) {
        --i;
        --j;
      } else if (t == 1) {
        --i;
      } else {
        --j;
      }
    }
    a.path_len[p] = cap - k;
  }
}

}  // namespace

hipError_t launch_dtw(const DtwArgs& a, hipStream_t s) {
  if (a.P <= 0) return hipSuccess;
  if (a.N_max <= 0 || a.M_max <= 0 || a.N_max > 512) return hipErrorInvalidValue;
  const int R = (a.N_max + 63) / 64;
  dim3 grid(a.P), block(64);
  switch (R) {
    case 1: hipLaunchKernelGGL((dtw_kernel<1, 8>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((dtw_kernel<2, 8>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((dtw_kernel<3, 4>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((dtw_kernel<4, 4>), grid, block, 0, s, a); break;
    case 5: hipLaunchKernelGGL((dtw_kernel<5, 2>), grid, block, 0, s, a); break;
    case 6: hipLaunchKernelGGL((dtw_kernel<6, 2>), grid, block, 0, s, a); break;
    case 7: hipLaunchKernelGGL((dtw_kernel<7, 2>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((dtw_kernel<8, 2>), grid, block, 0, s, a); break;
  }
  return hipGetLastError();
}

}  // namespace wca
