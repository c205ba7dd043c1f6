// Few-row GEMM of the greedy ASR pre-pass (one decode step: M = batch rows; reference call sites infer_ali.py:40,60-61
// `whisper.decode`) and of small teacher-forced forwards (M <= 128 rows, DEC_ROWS_MAX): C[m][n] = epilogue(sum_k A[m][k] * W[n][k]).
//
// These launches are latency-bound (a step is ~290 dependent kernels of 5-20 us), so the kernel is built to (1) take
// neighbouring small kernels INTO the GEMM and (2) have every weight byte in flight from the first instruction:
//   * A_MODE 1: the A operand is LayerNorm(x) of the fp32 residual rows, computed in the prologue (one wave per row, the
//     arithmetic of layernorm_f16_v4_kernel, bit-identical values) and kept as an f16 image in LDS -- the standalone
//     LayerNorm launch in front of the QKV / cross-query / fc1 / logits GEMMs disappears;
//     A_MODE 0: the f16 rows are copied into the same LDS image.
//   * the KV-cache append of the step (k / v columns of the QKV projection) is a routing rule of the f16 epilogue;
//   * split-K over `S` workgroups for K = 4 d (fc2, only N / 16 = 64 column groups otherwise): partial tiles to a
//     workspace with agent-scope (write-through) stores, the LAST workgroup to arrive (one atomic counter per column
//     group) reads them back with agent-scope loads (the XCDs' L2s are not coherent with each other; a fence would write
//     the whole L2 back: 30 us) and adds the S partials in fixed order -> deterministic;
//   * a workgroup's weight slice (16 columns x K/S) is requested in full before the prologue starts and lands under it;
//     with G > 1 column groups per workgroup (the vocabulary projection) a 3-deep register ring keeps two groups ahead.
// Workgroup = 8 waves. Prologue: wave w owns rows w, w + 8, ... and has all of its row loads in flight at once (one L2
// round trip for the image). Main loop: wave (kq, mp) multiplies K quarter kq of the slice for m-tiles 2 mp, 2 mp + 1 on
// v_mfma_f32_16x16x32_f16 with W as the A operand; the four quarter accumulators of an m-tile are summed through LDS in
// quarter order by waves 0..3 (the summation order of gemm_skinny_f16_kernel: S = 1 results are bit-identical to
// LayerNorm kernel + that kernel).
#include <atomic>

#include "kernels.h"
#include "wca_common.h"

namespace wca {

namespace {

template <int V>
struct IntC {
  static constexpr int value = V;
};

constexpr int ROWS = 64;     // rows of one workgroup (grid.y walks 64-row blocks)
constexpr int MAXU = 8;      // 32-wide k steps per wave: K / S <= 1024
constexpr int APAD = 8;      // halfs of padding per LDS row (row stride = 16 B mod 256 B: ds_read_b128 groups spread over the bank row)

// wave_sum (wca_common.h) on eight values at once: the same operations per value, stage by stage across the eight
__device__ __forceinline__ void wave_sum8(float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] += dpp_mov<0x121>(v[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] += dpp_mov<0x122>(v[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] += dpp_mov<0x124>(v[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] += dpp_mov<0x128>(v[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
    v[i] = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
    v[i] = __uint_as_float(b[0]) + __uint_as_float(b[1]);
  }
}

typedef unsigned long long __attribute__((address_space(1))) gu64_t;
typedef unsigned __attribute__((address_space(1))) gu32_t;

template <int A_MODE, int OUT_MODE, bool GELU>
__global__ __launch_bounds__(512) void gemm_rows_f16_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  const int kq = wave & 3;   // K quarter of the slice this wave multiplies
  const int mp = wave >> 2;  // ... for m-tiles 2 mp and 2 mp + 1
  const int fr = lane & 15, fg = lane >> 4;
  const int S = a.splitk;
  const int G = a.groups;
  const int ks = blockIdx.x % S;
  const int gblock = blockIdx.x / S;
  const int Ks = a.K / S;              // this workgroup's K slice
  const int kslo = ks * Ks;
  const int Kw = Ks >> 2;              // per K quarter
  const int nu = Kw >> 5;              // 32-wide steps per wave (<= MAXU)
  const int m_base = blockIdx.y * ROWS;
  const int rows = min(ROWS, a.M - m_base);
  const int mt_n = (rows + 15) >> 4;
  const int NG = (a.N + 15) >> 4;
  const int lds_ld = Ks + APAD;        // halfs
  half_t* As = reinterpret_cast<half_t*>(smem);
  float* red = reinterpret_cast<float*>(smem + (size_t)ROWS * lds_ld * sizeof(half_t));  // [4 kq][4 mt][4][64]
  int* flag = reinterpret_cast<int*>(red + 4 * 4 * 4 * 64);

  // ---- weight prefetch: group slot ring, everything of a group's K quarter for this wave in registers (the two waves of a
  // quarter request the same lines)
  half8 wf[3][MAXU];
  auto load_w = [&](auto slot_c, int cg) {
    constexpr int SL = decltype(slot_c)::value;
    int nrow = cg * 16 + fr;
    nrow = nrow < a.N ? nrow : a.N - 1;  // columns past N: duplicated weights, results not stored
    const half_t* wp = a.W + (long)nrow * a.ldw + kslo + kq * Kw + fg * 8;
#pragma unroll
    for (int u = 0; u < MAXU; ++u)
      if (u < nu) wf[SL][u] = *reinterpret_cast<const half8*>(wp + u * 32);
  };
  const int cg0 = gblock * G;
  load_w(IntC<0>{}, cg0);
  if (G > 1 && cg0 + 1 < NG) load_w(IntC<1>{}, cg0 + 1);

  // ---- prologue: the A image [rows][Ks] f16 in LDS. Wave w owns rows w, w + 8, ...: ALL of a wave's row loads are issued
  // before the first is consumed (one L2 round trip for the whole image, not one per row)
  if (A_MODE == 1) {
    const int NV = a.K >> 8;  // 16-byte vectors per lane of one row (K = 256 NV <= 1024)
    f32x4 g4[4], b4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < NV) {
        g4[i] = reinterpret_cast<const f32x4*>(a.ln_gamma)[i * 64 + lane];
        b4[i] = reinterpret_cast<const f32x4*>(a.ln_beta)[i * 64 + lane];
      }
    const float inv_d = 1.0f / (float)a.K;
    f32x4 v[8][4];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = wave + 8 * rr;
      const f32x4* xr = reinterpret_cast<const f32x4*>(a.A32 + (long)(m_base + (r < rows ? r : rows - 1)) * a.lda32);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < NV) v[rr][i] = xr[i * 64 + lane];
    }
    // the eight rows' reductions advance in lockstep (eight independent dependency chains per stage)
    float mean[8], rstd[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < NV) s += (v[rr][i][0] + v[rr][i][1]) + (v[rr][i][2] + v[rr][i][3]);
      mean[rr] = s;
    }
    wave_sum8(mean);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      mean[rr] *= inv_d;
      float q = 0.f;
      {
        // products rounded before they are added, as in layernorm_f16_v4_kernel (packed multiplies there): the two
        // kernels give the same statistics bit for bit
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < NV) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float t = v[rr][i][j] - mean[rr];
              const float tt = t * t;
              q += tt;
            }
          }
      }
      rstd[rr] = q;
    }
    wave_sum8(rstd);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = wave + 8 * rr;
      const float rs = rsqrtf(rstd[rr] * inv_d + a.ln_eps);
      if (r < rows) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < NV) {
            const int k = i * 256 + lane * 4 - kslo;
            half4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (half_t)((v[rr][i][j] - mean[rr]) * rs * g4[i][j] + b4[i][j]);
            if (k >= 0 && k < Ks) *reinterpret_cast<half4*>(As + (long)r * lds_ld + k) = o;
          }
      }
    }
  } else {
    const int cpr = Ks >> 3;  // 16-byte chunks per row (<= 128)
    half8 t[8][2];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = wave + 8 * rr;
      if (r < rows) {
        const half_t* src = a.A + (long)(m_base + r) * a.lda + kslo;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (lane + 64 * j < cpr) t[rr][j] = *reinterpret_cast<const half8*>(src + (lane + 64 * j) * 8);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = wave + 8 * rr;
      if (r < rows) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (lane + 64 * j < cpr) *reinterpret_cast<half8*>(As + (long)r * lds_ld + (lane + 64 * j) * 8) = t[rr][j];
      }
    }
  }
  __syncthreads();

  const half_t* a_lane = As + (long)fr * lds_ld + kq * Kw + fg * 8;
  const unsigned kv_d = (unsigned)a.kv_d;

  auto group = [&](auto slot_c, int gi) {
    constexpr int SL = decltype(slot_c)::value;
    const int cg = cg0 + gi;
    if (G > 1 && gi + 2 < G && cg + 2 < NG) load_w(IntC<(SL + 2) % 3>{}, cg + 2);
    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < MAXU; ++u)
      if (u < nu) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (mp * 2 + j < mt_n) {
            const half8 xa = *reinterpret_cast<const half8*>(a_lane + (long)(mp * 2 + j) * 16 * lds_ld + u * 32);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[SL][u], xa, acc[j], 0, 0, 0);
          }
      }
    if (gi > 0) __syncthreads();  // the previous group's `red` has been consumed
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (mp * 2 + j < mt_n) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((kq * 4 + mp * 2 + j) * 4 + r) * 64 + lane] = acc[j][r];
      }
    __syncthreads();
    // wave w < 4 finishes m-tile w (quarters added in order 0..3): lane holds C[m = w*16 + fr][n = 16 cg + 4 fg + r]
    const int ml = wave * 16 + fr;
    const int m = m_base + ml;
    const int nb = cg * 16 + 4 * fg;
    float v[4];
    const bool mine = wave < mt_n;  // mt_n <= 4
    if (mine) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[r] = red[((0 * 4 + wave) * 4 + r) * 64 + lane] + red[((1 * 4 + wave) * 4 + r) * 64 + lane] +
               red[((2 * 4 + wave) * 4 + r) * 64 + lane] + red[((3 * 4 + wave) * 4 + r) * 64 + lane];
    }
    if (S > 1) {
      // ---- split-K hand-off (no fences: an agent-scope fence writes the whole L2 back). Partial tile -> workspace with
      // agent-scope (write-through, sc1) stores, drained by every storing wave before the workgroup barrier; one lane counts
      // the arrival; the LAST of the S workgroups reads all S partials back with agent-scope loads and adds them in order.
      const long tile = (long)blockIdx.y * NG + cg;
      float* part = a.sk_part + (tile * S) * (ROWS * 16);
      if (mine) {
        gu64_t* dst = (gu64_t*)(part + (long)ks * (ROWS * 16) + ml * 16 + 4 * fg);
        __hip_atomic_store(dst, ((unsigned long long)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, ((unsigned long long)__float_as_uint(v[3]) << 32) | __float_as_uint(v[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wait_vm0();
      }
      __syncthreads();
      if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add((gu32_t*)(a.sk_cnt + tile), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = (old == (unsigned)(S - 1));
        if (old == (unsigned)(S - 1)) __hip_atomic_store((gu32_t*)(a.sk_cnt + tile), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // self-cleaning
      }
      __syncthreads();
      if (!*flag) return;
      if (mine) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = 0.f;
        for (int s = 0; s < S; ++s) {
          gu64_t* src = (gu64_t*)(part + (long)s * (ROWS * 16) + ml * 16 + 4 * fg);
          const unsigned long long p0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long p1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v[0] += __uint_as_float((unsigned)p0);
          v[1] += __uint_as_float((unsigned)(p0 >> 32));
          v[2] += __uint_as_float((unsigned)p1);
          v[3] += __uint_as_float((unsigned)(p1 >> 32));
        }
      }
    }
    if (!mine || ml >= rows) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nb + r;
      if (a.bias != nullptr && n < a.N) v[r] += a.bias[n];
      if (GELU) v[r] = gelu_erf(v[r]);
    }
    long coff;
    if (a.c_rows_per_batch > 0) {
      const int b = m / a.c_rows_per_batch;
      coff = (long)b * a.c_batch_stride + (long)(m - b * a.c_rows_per_batch) * a.ldc;
    } else {
      coff = (long)m * a.ldc;
    }
    if (OUT_MODE == 0) {
      half_t* cp = reinterpret_cast<half_t*>(a.C) + coff + nb;
      if (a.kv_k != nullptr && (unsigned)nb >= kv_d) {
        // KV-cache append: columns [d, 2d) -> K cache row, [2d, 3d) -> V cache row of batch row m at position kv_t
        const bool is_v = (unsigned)nb >= 2 * kv_d;
        cp = (is_v ? a.kv_v : a.kv_k) + (long)m * a.kv_bs + (long)a.kv_t * a.kv_d + (nb - (is_v ? 2 : 1) * a.kv_d);
      }
      if (nb + 4 <= a.N && (reinterpret_cast<uintptr_t>(cp) & 7) == 0) {
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
        *reinterpret_cast<half4*>(cp) = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nb + r < a.N) cp[r] = (half_t)v[r];
      }
    } else {
      float* cp = reinterpret_cast<float*>(a.C) + coff + nb;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nb + r < a.N) cp[r] = (OUT_MODE == 2) ? cp[r] + v[r] : v[r];
    }
  };

  // groups walk the register ring in order 0, 1, 2, 0, ... (static slots: the fragments must stay in registers)
  for (int g3 = 0; g3 < G; g3 += 3) {
    if (cg0 + g3 >= NG) break;
    group(IntC<0>{}, g3);
    if (g3 + 1 >= G || cg0 + g3 + 1 >= NG) break;
    group(IntC<1>{}, g3 + 1);
    if (g3 + 2 >= G || cg0 + g3 + 2 >= NG) break;
    group(IntC<2>{}, g3 + 2);
  }
}

}  // namespace

size_t gemm_rows_workspace_bytes(int M, int N, int splitk) {
  if (splitk <= 1) return 0;
  const size_t tiles = (size_t)((M + ROWS - 1) / ROWS) * ((N + 15) / 16);
  return tiles * splitk * ROWS * 16 * sizeof(float);
}

// Shapes this kernel takes: any M (64-row blocks), K a multiple of 128 * splitk with K / splitk <= 1024; with LayerNorm
// K = 256 NV <= 1024 and the fp32 rows 16-byte aligned. `splitk` 0 = chosen here.
int gemm_rows_pick_splitk(int K) {
  for (int s = 1; s <= 8; ++s)
    if (K % (s * 128) == 0 && K / s <= 1024) return s;
  return 0;
}

bool gemm_rows_supported(int M, int N, int K, bool layernorm_a) {
  if (M < 1 || N < 1 || gemm_rows_pick_splitk(K) == 0) return false;
  if (layernorm_a && ((K % 256) != 0 || K > 1024)) return false;
  return true;
}

hipError_t launch_gemm_rows(const GemmArgs& a_in, hipStream_t s) {
  GemmArgs a = a_in;
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (a.K <= 0 || a.a_rows_per_batch != 0 || a.pos != nullptr || a.addend != nullptr) return hipErrorInvalidValue;
  if (a.splitk <= 0) a.splitk = gemm_rows_pick_splitk(a.K);
  if (a.splitk <= 0 || a.K % (a.splitk * 128) != 0 || a.K / a.splitk > 1024) return hipErrorInvalidValue;
  if ((a.ldw % 8) != 0) return hipErrorInvalidValue;
  const bool ln = a.A32 != nullptr;
  if (ln) {
    if ((a.K % 256) != 0 || a.K > 1024 || (a.lda32 % 4) != 0 || !a.ln_gamma || !a.ln_beta) return hipErrorInvalidValue;
  } else {
    if (!a.A || (a.lda % 8) != 0) return hipErrorInvalidValue;
  }
  if (a.kv_k != nullptr && (a.out_mode != 0 || !a.kv_v || a.kv_d <= 0 || (a.kv_d % 16) != 0 || a.N != 3 * a.kv_d || a.c_rows_per_batch != 0))
    return hipErrorInvalidValue;
  const int NG = (a.N + 15) / 16;
  if (a.splitk > 1) {
    if (!a.sk_part || !a.sk_cnt) return hipErrorInvalidValue;
    a.groups = 1;
  } else if (a.groups <= 0) {
    a.groups = (NG + 255) / 256;
  }
  const dim3 grid((unsigned)(((NG + a.groups - 1) / a.groups) * a.splitk), (unsigned)((a.M + ROWS - 1) / ROWS)), block(512);
  const size_t shmem = (size_t)ROWS * (a.K / a.splitk + APAD) * sizeof(half_t) + 4 * 4 * 4 * 64 * sizeof(float) + 16;
  int dev = 0;
  {
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
  }
#define WCA_ROWS_K(AM, OM, G)                                                                                       \
  do {                                                                                                              \
    static std::atomic<unsigned> attr_mask{0};                                                                      \
    if (!(attr_mask.load(std::memory_order_acquire) & (1u << (dev & 31)))) {                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_f16_kernel<AM, OM, G>),            \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                   \
      if (e != hipSuccess) return e;                                                                                \
      attr_mask.fetch_or(1u << (dev & 31), std::memory_order_release);                                              \
    }                                                                                                               \
    hipLaunchKernelGGL((gemm_rows_f16_kernel<AM, OM, G>), grid, block, shmem, s, a);                                \
  } while (0)
#define WCA_ROWS_A(OM, G)                \
  do {                                   \
    if (ln) WCA_ROWS_K(1, OM, G);        \
    else WCA_ROWS_K(0, OM, G);           \
  } while (0)
  if (a.out_mode == 0) {
    if (a.gelu) WCA_ROWS_A(0, true); else WCA_ROWS_A(0, false);
  } else if (a.out_mode == 1 && !a.gelu) {
    WCA_ROWS_A(1, false);
  } else if (a.out_mode == 2 && !a.gelu) {
    WCA_ROWS_A(2, false);
  } else {
    return hipErrorInvalidValue;
  }
#undef WCA_ROWS_A
#undef WCA_ROWS_K
  return hipGetLastError();
}

}  // namespace wca
