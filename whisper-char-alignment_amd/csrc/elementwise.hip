// Row-wise small ops of the Whisper forward (HBM-bound, one wave per row, vectorised I/O).
//   layernorm_f16 : whisper.model.LayerNorm (fp32 statistics, eps 1e-5) feeding an f16 GEMM operand
//   embed         : token_embedding[tokens] + positional_embedding[:n]   (TextDecoder.forward)
#include <cstdlib>

#include "kernels.h"
#include "wca_common.h"

namespace wca {

namespace {

// One wave per row, the whole row in registers (two passes over registers, one over HBM).
// NV = 16-byte vectors per lane: d = 256 * NV (d = 256 .. 1280); d = 128 / 384 use the 8-byte variant.
// SPLIT: the output row (ld_out elements apart) carries hi = f16(y) at [0, d) and lo = f16(y - hi) at [lo_off, lo_off + d)
template <int NV, bool SPLIT = false>
__global__ __launch_bounds__(256) void layernorm_f16_v4_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, half_t* __restrict__ out,
                                                               int rows, float eps, int ld_out, long lo_off) {
  constexpr int d = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + (long)row * d);
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = xr[i * 64 + lane];
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) * (1.0f / d);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = v[i][j] - mean;
      q += t * t;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / d) + eps);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
  half4* o4 = reinterpret_cast<half4*>(out + (long)row * ld_out);
  half4* l4 = reinterpret_cast<half4*>(out + (long)row * ld_out + lo_off);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const f32x4 g = g4[i * 64 + lane], bb = b4[i * 64 + lane];
    half4 o, ol;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float y = (v[i][j] - mean) * rstd * g[j] + bb[j];
      if (SPLIT) {
        const HalfPair pr = split_pair(y);
        o[j] = pr.hi;
        ol[j] = pr.lo;
      } else {
        o[j] = (half_t)y;
      }
    }
    o4[i * 64 + lane] = o;
    if (SPLIT) l4[i * 64 + lane] = ol;
  }
}

// Pair output (SPLIT sites) with 16-byte stores: a lane owns EIGHT consecutive elements per 512-element chunk (two adjacent 16-byte
// loads, one half8 store each for the hi and the lo row half) instead of four -- the pair LayerNorm writes as many bytes as it reads,
// and the 8-byte stores of the kernel above held it at 4.4 TB/s (0.177 ms per 96000 x 1024 launch, 49 launches = 4.3 % of the
// contract-mode step). An odd NV leaves one 256-element chunk in the four-wide mapping. Same arithmetic; the fp32 summation order of
// the row statistics differs from the single-output kernel's (whose order gemm_rows.hip's LayerNorm prologue reproduces bit for bit).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_pair_v8_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, half_t* __restrict__ out, int rows, float eps,
                                                                int ld_out, long lo_off) {
  constexpr int d = 256 * NV, NC = NV / 2;  // NC chunks of 512 elements (+ one of 256 when NV is odd)
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * d;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    v[2 * c] = *reinterpret_cast<const f32x4*>(xr + c * 512 + lane * 8);
    v[2 * c + 1] = *reinterpret_cast<const f32x4*>(xr + c * 512 + lane * 8 + 4);
  }
  if (NV & 1) v[NV - 1] = *reinterpret_cast<const f32x4*>(xr + NC * 512 + lane * 4);
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum(s) * (1.0f / d);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = v[i][j] - mean;
      q += t * t;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / d) + eps);
  half_t* orow = out + (long)row * ld_out;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int e0 = c * 512 + lane * 8;
    half8 o, ol;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + e0 + 4 * h), bb = *reinterpret_cast<const f32x4*>(beta + e0 + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const HalfPair pr = split_pair((v[2 * c + h][j] - mean) * rstd * g[j] + bb[j]);
        o[4 * h + j] = pr.hi;
        ol[4 * h + j] = pr.lo;
      }
    }
    *reinterpret_cast<half8*>(orow + e0) = o;
    *reinterpret_cast<half8*>(orow + lo_off + e0) = ol;
  }
  if (NV & 1) {
    const int e0 = NC * 512 + lane * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + e0), bb = *reinterpret_cast<const f32x4*>(beta + e0);
    half4 o, ol;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const HalfPair pr = split_pair((v[NV - 1][j] - mean) * rstd * g[j] + bb[j]);
      o[j] = pr.hi;
      ol[j] = pr.lo;
    }
    *reinterpret_cast<half4*>(orow + e0) = o;
    *reinterpret_cast<half4*>(orow + lo_off + e0) = ol;
  }
}

template <int NV2, bool SPLIT = false>  // d = 128 * NV2
__global__ __launch_bounds__(256) void layernorm_f16_v2_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, half_t* __restrict__ out,
                                                               int rows, float eps, int ld_out, long lo_off) {
  constexpr int d = 128 * NV2;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x2* xr = reinterpret_cast<const f32x2*>(x + (long)row * d);
  f32x2 v[NV2];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV2; ++i) {
    v[i] = xr[i * 64 + lane];
    s += v[i][0] + v[i][1];
  }
  const float mean = wave_sum(s) * (1.0f / d);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV2; ++i) {
    const float a = v[i][0] - mean, b = v[i][1] - mean;
    q += a * a + b * b;
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / d) + eps);
  const f32x2* g2 = reinterpret_cast<const f32x2*>(gamma);
  const f32x2* b2 = reinterpret_cast<const f32x2*>(beta);
  half2_* o2 = reinterpret_cast<half2_*>(out + (long)row * ld_out);
  half2_* l2 = reinterpret_cast<half2_*>(out + (long)row * ld_out + lo_off);
#pragma unroll
  for (int i = 0; i < NV2; ++i) {
    const f32x2 g = g2[i * 64 + lane], bb = b2[i * 64 + lane];
    const float y0 = (v[i][0] - mean) * rstd * g[0] + bb[0], y1 = (v[i][1] - mean) * rstd * g[1] + bb[1];
    half2_ o;
    if (SPLIT) {
      const HalfPair p0 = split_pair(y0), p1 = split_pair(y1);
      half2_ ol;
      o[0] = p0.hi;
      o[1] = p1.hi;
      ol[0] = p0.lo;
      ol[1] = p1.lo;
      l2[i * 64 + lane] = ol;
    } else {
      o[0] = (half_t)y0;
      o[1] = (half_t)y1;
    }
    o2[i * 64 + lane] = o;
  }
}

// A token id outside [0, n_vocab) (tokenizer / checkpoint mismatch) must not become an out-of-bounds read of the
// embedding table: the row is embedded as token 0 and *err is raised; the host reports WCA_ERR_INVALID at its next sync.
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ tokens, const half_t* __restrict__ tok_emb,
                                                    const float* __restrict__ pos_emb, float* __restrict__ x, int B, int n,
                                                    int d, int n_vocab, int* __restrict__ err, const half_t* __restrict__ tok_emb_lo) {
  const int row = blockIdx.x;  // b*n + i
  const int i = row % n;
  long tok = tokens[row];
  if (tok < 0 || tok >= n_vocab) {
    if (threadIdx.x == 0 && err) atomicOr(err, 1);
    tok = 0;
  }
  const half_t* e = tok_emb + tok * d;
  const float* p = pos_emb + (long)i * d;
  float* o = x + (long)row * d;
  if (tok_emb_lo != nullptr) {   // an embedding table that is not exact in f16: value = hi + lo (engine.hip, W_lo slab)
    const half_t* el = tok_emb_lo + tok * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) o[c] = ((float)e[c] + (float)el[c]) + p[c];
    return;
  }
  for (int c = threadIdx.x; c < d; c += blockDim.x) o[c] = (float)e[c] + p[c];
}

__global__ void fill_f16_kernel(half_t* p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  const half_t h = (half_t)v;
  for (; i < n; i += st) p[i] = h;
}

__global__ void f32_to_f16_kernel(const float* __restrict__ in, half_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) out[i] = (half_t)in[i];
}

}  // namespace

hipError_t launch_layernorm_f16(const float* x, const float* gamma, const float* beta, half_t* out, int rows, int d,
                                float eps, hipStream_t s, int ld_out, long lo_off) {
  if (rows <= 0) return hipSuccess;
  if (ld_out <= 0) ld_out = d;
  if ((ld_out & 3) || (lo_off & 3) || lo_off < 0) return hipErrorInvalidValue;
  dim3 grid((rows + 3) / 4), block(256);
#define WCA_LN(KERN, NV)                                                                                                           \
  do {                                                                                                                             \
    if (lo_off) hipLaunchKernelGGL((KERN<NV, true>), grid, block, 0, s, x, gamma, beta, out, rows, eps, ld_out, lo_off);           \
    else hipLaunchKernelGGL((KERN<NV, false>), grid, block, 0, s, x, gamma, beta, out, rows, eps, ld_out, lo_off);                 \
  } while (0)
  // pair output on rows of >= 512 elements with 16-byte aligned halves: the eight-elements-per-lane kernel (16-byte stores)
  const bool pair_v8 = debug_switch(DBG_LN_PAIR_V4) == 0;   // (switch ln_pair_v4: the four-wide kernel, for the A/B)
  if (lo_off && pair_v8 && d >= 512 && (d % 256) == 0 && (ld_out & 7) == 0 && (lo_off & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    switch (d) {
      case 512: hipLaunchKernelGGL((layernorm_pair_v8_kernel<2>), grid, block, 0, s, x, gamma, beta, out, rows, eps, ld_out, lo_off); return hipGetLastError();
      case 768: hipLaunchKernelGGL((layernorm_pair_v8_kernel<3>), grid, block, 0, s, x, gamma, beta, out, rows, eps, ld_out, lo_off); return hipGetLastError();
      case 1024: hipLaunchKernelGGL((layernorm_pair_v8_kernel<4>), grid, block, 0, s, x, gamma, beta, out, rows, eps, ld_out, lo_off); return hipGetLastError();
      case 1280: hipLaunchKernelGGL((layernorm_pair_v8_kernel<5>), grid, block, 0, s, x, gamma, beta, out, rows, eps, ld_out, lo_off); return hipGetLastError();
      default: break;
    }
  }
  switch (d) {
    case 128: WCA_LN(layernorm_f16_v2_kernel, 1); break;
    case 256: WCA_LN(layernorm_f16_v4_kernel, 1); break;
    case 384: WCA_LN(layernorm_f16_v2_kernel, 3); break;
    case 512: WCA_LN(layernorm_f16_v4_kernel, 2); break;
    case 768: WCA_LN(layernorm_f16_v4_kernel, 3); break;
    case 1024: WCA_LN(layernorm_f16_v4_kernel, 4); break;
    case 1280: WCA_LN(layernorm_f16_v4_kernel, 5); break;
    default: return hipErrorInvalidValue;
  }
#undef WCA_LN
  return hipGetLastError();
}

hipError_t launch_embed(const int64_t* tokens, const half_t* tok_emb, const float* pos_emb, float* x, int B, int n,
                        int d, int n_vocab, int* err, hipStream_t s, const half_t* tok_emb_lo) {
  if (B * n <= 0) return hipSuccess;
  hipLaunchKernelGGL(embed_kernel, dim3(B * n), dim3(256), 0, s, tokens, tok_emb, pos_emb, x, B, n, d, n_vocab, err, tok_emb_lo);
  return hipGetLastError();
}

hipError_t launch_fill_f16(half_t* p, size_t n, float v, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(fill_f16_kernel, dim3(blocks), dim3(256), 0, s, p, n, v);
  return hipGetLastError();
}

hipError_t launch_f32_to_f16(const float* in, half_t* out, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3(blocks), dim3(256), 0, s, in, out, n);
  return hipGetLastError();
}

}  // namespace wca
