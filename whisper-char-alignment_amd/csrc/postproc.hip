// Post-processing of the captured cross-attention logits on gfx950 (all HBM-bound, no MFMA):
//   head_stats : median filter (reflect pad) -> *qk_scale -> softmax over frames      timing.py:64-66
//                + per-head score  w_col*sum_f||A[:,f]|| + w_row*sum_t||A[t,:]|| - w_cov*coverage   timing.py:13-34, metrics.py:99-111
//                in ONE pass over the logits (one workgroup per head, one wave per token row).
//   topk       : ascending top-k heads in python tuple order (score, (l,h))            timing.py:36
//   aggregate  : mean over selected heads of A / ||A||_col                              timing.py:84-97
//   median_filter : standalone whisper.timing.median_filter
#include <cstdlib>
#include "kernels.h"
#include "wca_common.h"

namespace wca {

namespace {

constexpr int MAX_NPL = 24;   // values per lane: frames <= 64 * 24 = 1536 >= 1500
constexpr int HALO = 16;      // supports filter widths up to 33

__device__ __forceinline__ void cswap(float& a, float& b) {
  const float lo = fminf(a, b), hi = fmaxf(a, b);
  a = lo;
  b = hi;
}

// median of W values starting at p (LDS), W odd, compile time
template <int W>
__device__ __forceinline__ float median_w(const float* p) {
  if (W == 1) return p[0];
  float v[W];
#pragma unroll
  for (int i = 0; i < W; ++i) v[i] = p[i];
  // odd-even transposition sort, W passes (branch free)
#pragma unroll
  for (int pass = 0; pass < W; ++pass) {
#pragma unroll
    for (int i = (pass & 1); i + 1 < W; i += 2) cswap(v[i], v[i + 1]);
  }
  return v[W / 2];
}

// generic odd width: rank selection by counting (ties broken by index) == sorted[w/2]
__device__ __forceinline__ float median_generic(const float* p, int w) {
  const int target = w >> 1;
  float res = p[0];
  for (int i = 0; i < w; ++i) {
    const float vi = p[i];
    int rank = 0;
    for (int j = 0; j < w; ++j) {
      const float vj = p[j];
      rank += (vj < vi) || (vj == vi && j < i);
    }
    if (rank == target) res = vi;
  }
  return res;
}

__device__ __forceinline__ float median_any(const float* p, int w) {
  switch (w) {
    case 1: return p[0];
    case 3: return median_w<3>(p);
    case 5: return median_w<5>(p);
    case 7: return median_w<7>(p);
    case 9: return median_w<9>(p);
    default: return median_generic(p, w);
  }
}

// Fill rowbuf[HALO + f] (f in [0,F)) from a global row, plus reflect halos of `pad` on both sides.
__device__ __forceinline__ void load_row_reflect(const float* __restrict__ src, float* rowbuf, int F, int pad, int lane) {
  for (int f = lane; f < F; f += 64) rowbuf[HALO + f] = src[f];
  __builtin_amdgcn_wave_barrier();
  if (lane < pad) {
    const int k = lane + 1;                       // 1..pad
    rowbuf[HALO - k] = rowbuf[HALO + k];           // x[-k] = x[k]
    rowbuf[HALO + F - 1 + k] = rowbuf[HALO + F - 1 - k];
  }
  __builtin_amdgcn_wave_barrier();
}

template <int NPL>
__global__ __launch_bounds__(256) void head_stats_kernel(HeadStatsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem);
  const int Fmax = a.n_frames_max;
  const int rb_stride = Fmax + 2 * HALO;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* rowbuf = sm + wave * rb_stride;
  float* red_sq = sm + 4 * rb_stride;            // [4][Fmax]
  float* red_sum = red_sq + 4 * Fmax;            // [4][Fmax]
  float* red_scalar = red_sum + 4 * Fmax;        // [16]

  const int head = blockIdx.x, b = blockIdx.y;
  const int n = a.n_tok[b], F = a.n_frames[b];
  const int w = a.medfilt_width;
  const int pad = w >> 1;
  const bool do_med = (w > 1) && (F > pad) && !a.input_is_weights;
  const float* qk = a.qk + (long)b * a.qk_bs + (long)head * a.qk_hs;
  float* wout = a.weights ? a.weights + (long)b * a.w_bs + (long)head * a.n_tok_max * Fmax : nullptr;

  float csq[NPL], csum[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    csq[i] = 0.f;
    csum[i] = 0.f;
  }
  float rown_acc = 0.f;

  // the next row of this wave is fetched into registers while the current one is processed
  float nxt[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int f = lane + 64 * i;
    nxt[i] = (wave < n && f < F) ? qk[(long)wave * a.qk_ld + f] : -INFINITY;
  }
  for (int t = wave; t < n; t += 4) {
    float v[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] = nxt[i];
    if (t + 4 < n) {
      const float* srcn = qk + (long)(t + 4) * a.qk_ld;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int f = lane + 64 * i;
        nxt[i] = (f < F) ? srcn[f] : -INFINITY;
      }
    }
    if (do_med) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int f = lane + 64 * i;
        if (f < F) rowbuf[HALO + f] = v[i];
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < pad) {
        const int k = lane + 1;
        rowbuf[HALO - k] = rowbuf[HALO + k];
        rowbuf[HALO + F - 1 + k] = rowbuf[HALO + F - 1 - k];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int f = lane + 64 * i;
        v[i] = (f < F) ? median_any(rowbuf + HALO + f - pad, w) : -INFINITY;
      }
      __builtin_amdgcn_wave_barrier();
    }
    float sum = 1.0f;
    if (!a.input_is_weights) {
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int f = lane + 64 * i;
        v[i] = (f < F) ? v[i] * a.qk_scale : -INFINITY;
        mx = fmaxf(mx, v[i]);
      }
      mx = wave_max(mx);
      sum = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int f = lane + 64 * i;
        v[i] = (f < F) ? expf(v[i] - mx) : 0.f;
        sum += v[i];
      }
      sum = wave_sum(sum);
      if (a.rowstats && lane == 0) {
        float* rsp = a.rowstats + (((long)b * a.LH + head) * a.n_tok_max + t) * 2;
        rsp[0] = mx;
        rsp[1] = sum;
      }
    }
    float rsq = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int f = lane + 64 * i;
      const float p = a.input_is_weights ? v[i] : v[i] / sum;
      if (f < F) {
        if (wout) wout[(long)t * Fmax + f] = p;
        rsq += p * p;
        csq[i] += p * p;
        csum[i] += p;
      }
    }
    rown_acc += sqrtf(wave_sum(rsq));
  }

  // ---- deterministic cross-wave reduction of the per-column statistics
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int f = lane + 64 * i;
    if (f < F) {
      red_sq[wave * Fmax + f] = csq[i];
      red_sum[wave * Fmax + f] = csum[i];
    }
  }
  if (lane == 0) red_scalar[wave] = rown_acc;
  __syncthreads();
  float cn_part = 0.f, cov_part = 0.f;
  float* cn_out = a.colnorm + ((long)b * a.LH + head) * Fmax;
  for (int f = tid; f < F; f += 256) {
    const float sq = ((red_sq[f] + red_sq[Fmax + f]) + red_sq[2 * Fmax + f]) + red_sq[3 * Fmax + f];
    const float sm_ = ((red_sum[f] + red_sum[Fmax + f]) + red_sum[2 * Fmax + f]) + red_sum[3 * Fmax + f];
    const float cn = sqrtf(sq);
    cn_out[f] = cn;
    cn_part += cn;
    cov_part += fmaxf(sm_, 0.5f);
  }
  cn_part = wave_sum(cn_part);
  cov_part = wave_sum(cov_part);
  if (lane == 0) {
    red_scalar[4 + wave] = cn_part;
    red_scalar[8 + wave] = cov_part;
  }
  __syncthreads();
  if (tid == 0) {
    const float rown = ((red_scalar[0] + red_scalar[1]) + red_scalar[2]) + red_scalar[3];
    const float coln = ((red_scalar[4] + red_scalar[5]) + red_scalar[6]) + red_scalar[7];
    const float cov = (((red_scalar[8] + red_scalar[9]) + red_scalar[10]) + red_scalar[11]) - 0.5f * (float)F;
    float score = 0.f;
    if (a.w_col > 0.f) score += a.w_col * coln;
    if (a.w_row > 0.f) score += a.w_row * rown;
    if (a.w_cov > 0.f) score -= a.w_cov * cov;
    a.scores[(long)b * a.LH + head] = score;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// head_stats_fast_kernel: the SAME arithmetic as head_stats_kernel, bit for bit (tests/test_postproc_gpu.py compares the two on every
// output), written for the vector-instruction budget: the general kernel above is bound by vector issue, not by HBM (2.9 TB/s: ~45 vector
// instructions and 5 exec-mask branches per logit). Here, for filter widths 1 / 3 / 5 / 7 / 9 and logits as input:
//   * no exec masking in the row loop: column offsets are clamped to F - 1 (every load is in bounds), lanes past F are set to -inf by ONE
//     v_cndmask after the median (their LDS garbage reaches no valid column: the reflect halo is written after the row);
//   * rows are addressed by a scalar base (wave index through readfirstlane) + a per-lane column offset: no 64-bit vector address math;
//   * the median of 3 is v_med3_f32 on (LDS left, register centre, LDS right), of 5 the 7-operation min / max / med3 form -- a median is
//     one of its inputs, so any correct selection gives the same bits as the sorting network of median_w;
//   * expf: libm's own sequence (ph = x log2e, pl its rounding error + x log2e_lo, e = rint(ph), ldexp(exp2((ph - e) + pl), e), 0 below
//     -103.28) on PAIRS of elements (v_pk_mul / v_pk_fma / v_pk_add_f32) without its overflow arm (x = v - max <= 0);
//   * p = e / sum: the reciprocal refinement of the IEEE division expansion depends on the row sum only -- done once per row; per element the
//     five remaining steps q0 = n r, e1 = fma(-d, q0, n), q1 = fma(e1, r, q0), e2 = fma(-d, q1, n), q = fma(e2, r, q1) as packed operations.
//     That IS v_div_scale / v_div_fmas / v_div_fixup's result whenever no operand scaling triggers (sum in [1, 1536]: numerator >= 2^-100
//     or 0); a row holding a smaller non-zero numerator (x < -68 on a valid column) takes the plain `/` instead;
//   * wave reductions with the DPP operand folded into the add / max (same summation tree as wave_sum).
__device__ __forceinline__ float vmin_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmin3_raw(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmed3_raw(float a, float b, float c) {
  float r;
  asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// wave_sum's tree (v += ror 1, 2, 4, 8 inside rows of 16, then the 16- and 32-lane swaps) with the rotate folded into the add
__device__ __forceinline__ float wave_sum_folded(float v) {
  asm volatile(
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
      : "+v"(v));
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_max_folded(float v) {
  asm volatile(
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
      : "+v"(v));
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = vmax_raw(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return vmax_raw(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// median of the W values centred on p[0] (LDS), `c` = p[0] already in a register
template <int W>
__device__ __forceinline__ float median_centre(const float* p, float c) {
  if (W == 1) return c;
  if (W == 3) return vmed3_raw(p[-1], c, p[1]);
  if (W == 5) {
    const float a = p[-2], b = p[-1], d = p[1], e = p[2];
    const float lo = vmax_raw(vmin_raw(a, b), vmin_raw(d, e)), hi = vmin_raw(vmax_raw(a, b), vmax_raw(d, e));
    return vmed3_raw(c, lo, hi);
  }
  // 7 and 9: median-selection networks (13 / 19 compare-exchanges; checked over all 0/1 inputs), 20 / 30 live min / max operations once the
  // exchanges whose results the median does not depend on are dropped (the unused halves are dead code) against the 42 / 72 of the full sort
  float q[W];
#pragma unroll
  for (int i = 0; i < W; ++i) q[i] = i == (W >> 1) ? c : p[i - (W >> 1)];
#define WCA_CX(a, b)                                  \
  {                                                   \
    const float lo_ = vmin_raw(q[a], q[b]);           \
    const float hi_ = vmax_raw(q[a], q[b]);           \
    q[a] = lo_;                                       \
    q[b] = hi_;                                       \
  }
  if (W == 7) {
    WCA_CX(0, 5) WCA_CX(0, 3) WCA_CX(1, 6) WCA_CX(2, 4) WCA_CX(0, 1) WCA_CX(3, 5) WCA_CX(2, 6) WCA_CX(2, 3) WCA_CX(3, 6) WCA_CX(4, 5)
    WCA_CX(1, 4) WCA_CX(1, 3) WCA_CX(3, 4)
    return q[3];
  }
  if (W == 9) {
    WCA_CX(1, 2) WCA_CX(4, 5) WCA_CX(7, 8) WCA_CX(0, 1) WCA_CX(3, 4) WCA_CX(6, 7) WCA_CX(1, 2) WCA_CX(4, 5) WCA_CX(7, 8) WCA_CX(0, 3)
    WCA_CX(5, 8) WCA_CX(4, 7) WCA_CX(3, 6) WCA_CX(1, 4) WCA_CX(2, 5) WCA_CX(4, 7) WCA_CX(4, 2) WCA_CX(6, 4) WCA_CX(4, 2)
    return q[4];
  }
#undef WCA_CX
  return median_w<W>(p - (W >> 1));
}

// expf(x) for x <= 0 (or NaN): the instruction sequence of the device libm's expf on two elements, overflow arm dropped
__device__ __forceinline__ f32x2 expf_nonpos2(f32x2 x) {
  const float L2E = __uint_as_float(0x3fb8aa3bu), L2E_LO = __uint_as_float(0x32a5705fu), UNDER = __uint_as_float(0xc2ce8ed0u);
  f32x2 ph = x * L2E;
  asm volatile("" : "+v"(ph));  // no contraction of this product into the subtractions below
  f32x2 pl = __builtin_elementwise_fma(x, (f32x2){L2E, L2E}, -ph);
  pl = __builtin_elementwise_fma(x, (f32x2){L2E_LO, L2E_LO}, pl);
  const f32x2 e = {__builtin_rintf(ph.x), __builtin_rintf(ph.y)};
  f32x2 a = ph - e;
  asm volatile("" : "+v"(a));
  a = a + pl;
  f32x2 r;
  r.x = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(a.x), (int)e.x);
  r.y = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(a.y), (int)e.y);
  r.x = (UNDER > x.x) ? 0.f : r.x;
  r.y = (UNDER > x.y) ? 0.f : r.y;
  return r;
}

template <int NPL, int W>
__global__ __launch_bounds__(256) void head_stats_fast_kernel(HeadStatsArgs a) {
  static_assert(NPL % 2 == 0, "pairs of elements");
  constexpr int NP = NPL / 2;
  constexpr int RB = 64 * NPL + 2 * HALO;        // every lane slot of a row has an LDS word (lanes past F write garbage there)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem);
  const int Fmax = a.n_frames_max;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* rowbuf = sm + wave * RB;
  float* red_sq = sm + 4 * RB;                   // [4][Fmax]
  float* red_sum = red_sq + 4 * Fmax;            // [4][Fmax]
  float* red_scalar = red_sum + 4 * Fmax;        // [16]

  const int head = blockIdx.x, b = blockIdx.y;
  const int n = a.n_tok[b], F = a.n_frames[b];
  constexpr int pad = W >> 1;
  const bool do_med = (W > 1) && (F > pad);
  const float* qk = a.qk + (long)b * a.qk_bs + (long)head * a.qk_hs;
  float* wout = a.weights ? a.weights + (long)b * a.w_bs + (long)head * a.n_tok_max * Fmax : nullptr;
  float* rsp = a.rowstats ? a.rowstats + ((long)b * a.LH + head) * a.n_tok_max * 2 : nullptr;
  const float scale = a.qk_scale;

  // rows through a buffer descriptor on this head's block: scalar row offset + 32-bit vector column offset, no vector address arithmetic
  const int row_bytes = a.qk_ld * 4;
  const long head_bytes = n > 0 && F > 0 ? ((long)(n - 1) * a.qk_ld + F) * 4 : 0;
  const __amdgpu_buffer_rsrc_t rq =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qk), 0, (int)(head_bytes < 0x7fffffffL ? head_bytes : 0x7fffffffL), 0x00020000);
  unsigned colofs[NPL];   // BYTE offset of the lane's column, clamped into the row: scalar row base + 32-bit vector offset addressing
  bool valid[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int f = lane + 64 * i;
    valid[i] = f < F;
    colofs[i] = 4u * (unsigned)max(min(f, F - 1), 0);
  }
  f32x2 csq[NP], csum[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    csq[j] = (f32x2){0.f, 0.f};
    csum[j] = (f32x2){0.f, 0.f};
  }
  float rown_acc = 0.f;

  float nxt[NPL];
  if (wave < n && F > 0) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) nxt[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rq, (int)colofs[i], wave * row_bytes, 0));
  } else {
#pragma unroll
    for (int i = 0; i < NPL; ++i) nxt[i] = 0.f;
  }
  if (F <= 0) {  // no frames: what the masked form leaves behind (row maximum -inf, sum 0, nothing accumulated)
    if (rsp && lane == 0)
      for (int t = wave; t < n; t += 4) {
        rsp[2 * t] = -INFINITY;
        rsp[2 * t + 1] = 0.f;
      }
  }
  for (int t = wave; t < n && F > 0; t += 4) {
    float v[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] = nxt[i];
    if (t + 4 < n) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) nxt[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rq, (int)colofs[i], (t + 4) * row_bytes, 0));
    }
    if (do_med) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) rowbuf[HALO + lane + 64 * i] = v[i];
      __builtin_amdgcn_wave_barrier();
      if (lane < pad) {
        const int k = lane + 1;
        rowbuf[HALO - k] = rowbuf[HALO + k];
        rowbuf[HALO + F - 1 + k] = rowbuf[HALO + F - 1 - k];
      }
      __builtin_amdgcn_wave_barrier();
      if (W == 3) {  // all neighbour reads in flight before the first median
        float lf[NPL], rt[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          lf[i] = rowbuf[HALO + lane + 64 * i - 1];
          rt[i] = rowbuf[HALO + lane + 64 * i + 1];
        }
#pragma unroll
        for (int i = 0; i < NPL; ++i) v[i] = __builtin_amdgcn_fmed3f(lf[i], v[i], rt[i]);
      } else {
#pragma unroll
        for (int i = 0; i < NPL; ++i) v[i] = median_centre<W>(rowbuf + HALO + lane + 64 * i, v[i]);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // scale; lanes past F -> -inf; row maximum and (over the finite values, garbage lanes included: conservative) row minimum
    f32x2 s2[NP];
    float mx = -INFINITY, mn = INFINITY;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      f32x2 sv = (f32x2){v[2 * j], v[2 * j + 1]} * scale;
      mn = vmin3_raw(mn, sv.x, sv.y);
      sv.x = valid[2 * j] ? sv.x : -INFINITY;
      sv.y = valid[2 * j + 1] ? sv.y : -INFINITY;
      mx = vmax3_raw(mx, sv.x, sv.y);
      s2[j] = sv;
    }
    mx = wave_max_folded(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      s2[j] = expf_nonpos2(s2[j] - mx);
      sum += s2[j].x;
      sum += s2[j].y;
    }
    sum = wave_sum_folded(sum);
    if (rsp && lane == 0) {
      rsp[2 * t] = mx;
      rsp[2 * t + 1] = sum;
    }
    // a non-zero numerator below 2^-100 makes the IEEE expansion scale its operands: those rows divide the plain way
    const bool tiny = __builtin_amdgcn_ballot_w64(mn - mx < -68.0f) != 0;
    float rsq = 0.f;
    if (!tiny) {
      const float r0 = __builtin_amdgcn_rcpf(sum);
      const float e0 = __builtin_fmaf(-sum, r0, 1.0f);
      const float r1 = __builtin_fmaf(e0, r0, r0);
      const f32x2 nd = {-sum, -sum}, rr = {r1, r1};
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const f32x2 nn = s2[j];
        f32x2 q = nn * rr;
        asm volatile("" : "+v"(q));
        f32x2 er = __builtin_elementwise_fma(nd, q, nn);
        q = __builtin_elementwise_fma(er, rr, q);
        er = __builtin_elementwise_fma(nd, q, nn);
        q = __builtin_elementwise_fma(er, rr, q);
        s2[j] = q;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        s2[j].x = s2[j].x / sum;
        s2[j].y = s2[j].y / sum;
      }
    }
    if (wout) {
      float* wrow = wout + (long)t * Fmax;
#pragma unroll
      for (int i = 0; i < NPL; ++i)
        if (valid[i]) wrow[lane + 64 * i] = (i & 1) ? s2[i >> 1].y : s2[i >> 1].x;
    }
    // lanes past F hold p = 0 (expf(-inf) = 0, 0 / sum = 0): they add nothing below, exactly like the masked form
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const f32x2 p = s2[j];
      rsq = __builtin_fmaf(p.x, p.x, rsq);
      rsq = __builtin_fmaf(p.y, p.y, rsq);
      csq[j] = __builtin_elementwise_fma(p, p, csq[j]);
      csum[j] = csum[j] + p;
    }
    rown_acc += sqrtf(wave_sum_folded(rsq));
  }

  // ---- deterministic cross-wave reduction of the per-column statistics (as in head_stats_kernel)
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int f = lane + 64 * i;
    if (f < F) {
      red_sq[wave * Fmax + f] = (i & 1) ? csq[i >> 1].y : csq[i >> 1].x;
      red_sum[wave * Fmax + f] = (i & 1) ? csum[i >> 1].y : csum[i >> 1].x;
    }
  }
  if (lane == 0) red_scalar[wave] = rown_acc;
  __syncthreads();
  float cn_part = 0.f, cov_part = 0.f;
  float* cn_out = a.colnorm + ((long)b * a.LH + head) * Fmax;
  for (int f = tid; f < F; f += 256) {
    const float sq = ((red_sq[f] + red_sq[Fmax + f]) + red_sq[2 * Fmax + f]) + red_sq[3 * Fmax + f];
    const float sm_ = ((red_sum[f] + red_sum[Fmax + f]) + red_sum[2 * Fmax + f]) + red_sum[3 * Fmax + f];
    const float cn = sqrtf(sq);
    cn_out[f] = cn;
    cn_part += cn;
    cov_part += fmaxf(sm_, 0.5f);
  }
  cn_part = wave_sum(cn_part);
  cov_part = wave_sum(cov_part);
  if (lane == 0) {
    red_scalar[4 + wave] = cn_part;
    red_scalar[8 + wave] = cov_part;
  }
  __syncthreads();
  if (tid == 0) {
    const float rown = ((red_scalar[0] + red_scalar[1]) + red_scalar[2]) + red_scalar[3];
    const float coln = ((red_scalar[4] + red_scalar[5]) + red_scalar[6]) + red_scalar[7];
    const float cov = (((red_scalar[8] + red_scalar[9]) + red_scalar[10]) + red_scalar[11]) - 0.5f * (float)F;
    float score = 0.f;
    if (a.w_col > 0.f) score += a.w_col * coln;
    if (a.w_row > 0.f) score += a.w_row * rown;
    if (a.w_cov > 0.f) score -= a.w_cov * cov;
    a.scores[(long)b * a.LH + head] = score;
  }
}

__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ scores, int LH, int k, int* __restrict__ sel_idx,
                                                   float* __restrict__ sel_score) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc = reinterpret_cast<float*>(smem);
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < LH; i += blockDim.x) sc[i] = scores[(long)b * LH + i];
  __syncthreads();
  const int keff = k < LH ? k : LH;
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    if (i >= keff) {
      sel_idx[(long)b * k + i] = -1;
      sel_score[(long)b * k + i] = 0.f;
    }
  }
  for (int i = threadIdx.x; i < LH; i += blockDim.x) {
    const float si = sc[i];
    int rank = 0;  // number of heads that sort AFTER head i in ascending (score, index) order
    for (int j = 0; j < LH; ++j) {
      const float sj = sc[j];
      rank += (sj > si) || (sj == si && j > i);
    }
    if (rank < keff) {
      sel_idx[(long)b * k + (keff - 1 - rank)] = i;
      sel_score[(long)b * k + (keff - 1 - rank)] = si;
    }
  }
}

// value of the filtered + softmaxed map of head `hd` at (t, f), recomputed from the captured logits exactly as
// head_stats_kernel computed it (same median, same expf(v - max) / sum), so both paths give identical bits
__device__ __forceinline__ float remat_weight(const AggregateArgs& a, int b, int hd, int t, int f, int F) {
  const float* row = a.qk + (long)b * a.qk_bs + (long)hd * a.qk_hs + (long)t * a.qk_ld;
  const float* rs = a.rowstats + (((long)b * a.LH + hd) * a.n_tok_max + t) * 2;
  const int w = a.medfilt_width, pad = w >> 1;
  float med;
  if (w <= 1 || F <= pad) {
    med = row[f];
  } else {
    float win[33];
    for (int k = 0; k < w; ++k) {
      int idx = f - pad + k;
      idx = idx < 0 ? -idx : idx;
      idx = idx >= F ? 2 * (F - 1) - idx : idx;
      win[k] = row[idx];
    }
    med = (w == 3) ? median_w<3>(win) : (w == 5) ? median_w<5>(win) : (w == 7) ? median_w<7>(win) : median_generic(win, w);
  }
  return expf(med * a.qk_scale - rs[0]) / rs[1];
}

__global__ __launch_bounds__(256) void aggregate_kernel(AggregateArgs a) {
  const int b = blockIdx.z;
  const int n = a.n_tok[b], F = a.n_frames[b];
  const int t = a.row_lo + blockIdx.y;
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (t >= n - a.row_hi_trim || f >= F) return;
  const int Fmax = a.n_frames_max;
  const float* W = a.weights ? a.weights + (long)b * a.w_bs : nullptr;
  const float* CN = a.colnorm + (long)b * a.LH * Fmax;
  float acc = 0.f;
  int cnt;
  auto contrib = [&](int hd) -> float {
    const float wv = W ? W[((long)hd * a.n_tok_max + t) * Fmax + f] : remat_weight(a, b, hd, t, f, F);
    return wv / CN[(long)hd * Fmax + f];
  };
  if (a.sel_idx) {
    cnt = 0;
    for (int s = 0; s < a.n_sel; ++s) {
      const int hd = a.sel_idx[(long)b * a.n_sel + s];
      if (hd < 0) continue;
      acc += contrib(hd);
      ++cnt;
    }
  } else {
    cnt = a.LH - a.head_lo;
    for (int hd = a.head_lo; hd < a.LH; ++hd) acc += contrib(hd);
  }
  a.matrix[((long)b * a.n_tok_max + (t - a.row_lo)) * Fmax + f] = acc / (float)cnt;
}

// default_find_alignment's normalisation (timing.py:159-160): per selected head and frame, (w - mean) / std over the
// token axis with the population std, two passes like torch.std_mean. One thread per (head, frame) column; the
// n <= 448 rows of a column are read three times (the second and third from L2). out [n_sel][n][F].
__global__ __launch_bounds__(256) void stdmean_normalize_kernel(const float* __restrict__ ws, const int* __restrict__ sel, int n, int F,
                                                                float* __restrict__ out) {
  const int s = blockIdx.y;
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const float* col = ws + (long)sel[s] * n * F + f;
  float sum = 0.f;
  for (int t = 0; t < n; ++t) sum += col[(long)t * F];
  const float mean = sum / (float)n;
  float sq = 0.f;
  for (int t = 0; t < n; ++t) {
    const float d = col[(long)t * F] - mean;
    sq = fmaf(d, d, sq);
  }
  const float sd = sqrtf(sq / (float)n);
  float* o = out + (long)s * n * F + f;
  for (int t = 0; t < n; ++t) o[(long)t * F] = (col[(long)t * F] - mean) / sd;
}

// matrix[t - row_lo][f] = mean over the n_sel heads of x[s][t][f], rows [row_lo, n - row_hi_trim)  (timing.py:162-163)
__global__ __launch_bounds__(256) void mean_heads_kernel(const float* __restrict__ x, int n_sel, int n, int F, int row_lo, int row_hi_trim,
                                                         float* __restrict__ matrix) {
  const int t = row_lo + blockIdx.y;
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (t >= n - row_hi_trim || f >= F) return;
  float acc = 0.f;
  for (int s = 0; s < n_sel; ++s) acc += x[((long)s * n + t) * F + f];
  matrix[(long)(t - row_lo) * F + f] = acc / (float)n_sel;
}

__global__ __launch_bounds__(256) void median_filter_kernel(const float* __restrict__ in, float* __restrict__ out, long rows,
                                                            int F, int w) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* rowbuf = sm + wave * (F + 2 * HALO);
  const int pad = w >> 1;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float* src = in + row * F;
    float* dst = out + row * F;
    if (w <= 1 || F <= pad) {
      for (int f = lane; f < F; f += 64) dst[f] = src[f];
      continue;
    }
    load_row_reflect(src, rowbuf, F, pad, lane);
    for (int f = lane; f < F; f += 64) dst[f] = median_any(rowbuf + HALO + f - pad, w);
    __builtin_amdgcn_wave_barrier();
  }
}

// probe_oracle.py:83-90 + metrics.py:45-72 (eval_n1_strict) for every head at once: head hd's predicted word boundaries are
// jump[hd][wb_end[i]] / 50 s; a prediction i counts if an UNUSED reference boundary j of the same word (eq[i][j]) lies within
// `tol` -- first such j in order, as the reference's double loop. One thread per head; times in double like the host code
// (frames / 50.0 compared in float64), so tp is the same integer.
__global__ __launch_bounds__(64) void probe_strict_kernel(const int* __restrict__ jump, int jump_ld, int LH, const int* __restrict__ wb_end,
                                                          int n_hyp, const double* __restrict__ y, int n_ref,
                                                          const unsigned char* __restrict__ eq, double tol, int* __restrict__ tp_out) {
  const int hd = blockIdx.x * blockDim.x + threadIdx.x;
  if (hd >= LH) return;
  unsigned long long used[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // n_ref <= 512
  int tp = 0;
  for (int i = 0; i < n_hyp; ++i) {
    const double yh = (double)jump[(long)hd * jump_ld + wb_end[i]] / 50.0;
    const unsigned char* e = eq + (long)i * n_ref;
    for (int j = 0; j < n_ref; ++j) {
      if (!e[j] || ((used[j >> 6] >> (j & 63)) & 1ull)) continue;
      if (fabs(y[j] - yh) <= tol) {
        used[j >> 6] |= 1ull << (j & 63);
        ++tp;
        break;
      }
    }
  }
  tp_out[hd] = tp;
}

}  // namespace

hipError_t launch_head_stats(const HeadStatsArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.LH <= 0) return hipSuccess;
  if (a.n_frames_max <= 0 || a.n_frames_max > 64 * MAX_NPL) return hipErrorInvalidValue;
  if (a.medfilt_width < 1 || (a.medfilt_width & 1) == 0 || (a.medfilt_width >> 1) > HALO) return hipErrorInvalidValue;
  const int Fmax = a.n_frames_max;
  dim3 grid(a.LH, a.B), block(256);
  const int w = a.medfilt_width;
  // the instruction-lean kernel (same bits) for logits input and the unrolled filter widths; the switch head_stats_general keeps the general one (A/B, tests)
  if (!a.input_is_weights && w <= 9 && !debug_switch(DBG_HEAD_STATS_GENERAL)) {
#define WCA_HSF(NPL, W)                                                                                   \
  do {                                                                                                    \
    const size_t shm = sizeof(float) * (4 * (size_t)(64 * NPL + 2 * HALO) + 8 * (size_t)Fmax + 16);      \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_stats_fast_kernel<NPL, W>),     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);             \
    if (e != hipSuccess) return e;                                                                        \
    hipLaunchKernelGGL((head_stats_fast_kernel<NPL, W>), grid, block, shm, s, a);                         \
    return hipGetLastError();                                                                             \
  } while (0)
#define WCA_HSF_W(NPL)                     \
  switch (w) {                             \
    case 1: WCA_HSF(NPL, 1);               \
    case 3: WCA_HSF(NPL, 3);               \
    case 5: WCA_HSF(NPL, 5);               \
    case 7: WCA_HSF(NPL, 7);               \
    default: WCA_HSF(NPL, 9);              \
  }
    if (Fmax <= 64 * 8) { WCA_HSF_W(8); }
    else if (Fmax <= 64 * 16) { WCA_HSF_W(16); }
    else { WCA_HSF_W(24); }
#undef WCA_HSF_W
#undef WCA_HSF
  }
  const size_t shmem = sizeof(float) * (4 * (size_t)(Fmax + 2 * HALO) + 8 * (size_t)Fmax + 16);
#define WCA_HS(NPL)                                                                                \
  do {                                                                                             \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_stats_kernel<NPL>),      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);    \
    if (e != hipSuccess) return e;                                                                 \
    hipLaunchKernelGGL((head_stats_kernel<NPL>), grid, block, shmem, s, a);                        \
  } while (0)
  if (Fmax <= 64 * 8) WCA_HS(8);
  else if (Fmax <= 64 * 16) WCA_HS(16);
  else WCA_HS(24);
#undef WCA_HS
  return hipGetLastError();
}

hipError_t launch_topk(const float* scores, int LH, int B, int k, int* sel_idx, float* sel_score, hipStream_t s) {
  if (B <= 0 || k <= 0) return hipSuccess;
  if (LH <= 0 || LH > 8192) return hipErrorInvalidValue;
  hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(256), sizeof(float) * LH, s, scores, LH, k, sel_idx, sel_score);
  return hipGetLastError();
}

hipError_t launch_aggregate(const AggregateArgs& a, hipStream_t s) {
  if (a.B <= 0) return hipSuccess;
  const int rows = a.n_tok_max - a.row_lo - a.row_hi_trim;
  if (rows <= 0) return hipSuccess;
  dim3 grid((a.n_frames_max + 255) / 256, rows, a.B), block(256);
  hipLaunchKernelGGL(aggregate_kernel, grid, block, 0, s, a);
  return hipGetLastError();
}

hipError_t launch_stdmean_normalize(const float* ws, const int* sel_dev, int n_sel, int n, int F, float* out, hipStream_t s) {
  if (n_sel <= 0 || n <= 0 || F <= 0) return hipSuccess;
  hipLaunchKernelGGL(stdmean_normalize_kernel, dim3((F + 255) / 256, n_sel), dim3(256), 0, s, ws, sel_dev, n, F, out);
  return hipGetLastError();
}

hipError_t launch_mean_heads(const float* x, int n_sel, int n, int F, int row_lo, int row_hi_trim, float* matrix, hipStream_t s) {
  const int rows = n - row_lo - row_hi_trim;
  if (n_sel <= 0 || rows <= 0 || F <= 0) return hipSuccess;
  hipLaunchKernelGGL(mean_heads_kernel, dim3((F + 255) / 256, rows), dim3(256), 0, s, x, n_sel, n, F, row_lo, row_hi_trim, matrix);
  return hipGetLastError();
}

hipError_t launch_probe_strict(const int* jump, int jump_ld, int LH, const int* wb_end, int n_hyp, const double* y, int n_ref,
                               const unsigned char* eq, double tol, int* tp_out, hipStream_t s) {
  if (LH <= 0) return hipSuccess;
  if (n_ref > 512 || n_ref < 0 || n_hyp < 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(probe_strict_kernel, dim3((LH + 63) / 64), dim3(64), 0, s, jump, jump_ld, LH, wb_end, n_hyp, y, n_ref, eq, tol, tp_out);
  return hipGetLastError();
}

hipError_t launch_median_filter(const float* in, float* out, long rows, int F, int width, hipStream_t s) {
  if (rows <= 0 || F <= 0) return hipSuccess;
  if (width < 1 || (width & 1) == 0 || (width >> 1) > HALO) return hipErrorInvalidValue;
  const size_t shmem = sizeof(float) * 4 * (size_t)(F + 2 * HALO);
  if (shmem > 160 * 1024) return hipErrorInvalidValue;
  long blocks = (rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(median_filter_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(median_filter_kernel, dim3((unsigned)blocks), dim3(256), shmem, s, in, out, rows, F, width);
  return hipGetLastError();
}

}  // namespace wca
