// Reference-precision ("split") attention for gfx950, head_dim 64: the attention of attention.hip's attn_kernel with every f16
// operand carried as a PAIR (hi = f16(x), lo = f16(x - hi), x = hi + lo to 2^-22) and each product as three MFMA passes:
//     S   = Qhi Khi^T + Qhi Klo^T + Qlo Khi^T          (the lo.lo term is below 2^-22 of the result and dropped)
//     O^T = Vhi^T Phi^T + Vhi^T Plo^T + Vlo^T Phi^T
// f16 x f16 products are exact in the MFMA's fp32 accumulator, so what remains is fp32 summation noise: the forward of
// timing.py:58 (`model(mel, tokens)` with fp32 parameters under disable_sdpa(): qk = (q*s)@(k*s)^T; softmax(qk.float()); @v) at
// fp32 accuracy on the f16 matrix pipe. Selected by AttnArgs.split (wca_set_precision(e, WCA_PRECISION_SPLIT)); used for the
// encoder self-attention, the decoder's causal self-attention and the cross-attention with CAPTURE of the pre-softmax logits
// (timing.py:50-55) -- the captured values are the three-pass fp32 sums times scale.
//
// Structure (deliberately the simple one; this mode trades speed for the reference's precision): 256-thread workgroup = 4 waves
// x 32 query rows, two workgroups per CU; 64-key tiles of K hi | K lo | V hi | V lo (4 x 8 KiB) in a ring of two LDS slots
// filled by LDS-DMA one tile ahead; one barrier per tile (at the top: tile kt has landed for every wave, and every wave is done
// with the slot that tile kt + 1 is then requested into). Scores transposed (S^T = K Q^T) so that a query row is a lane column;
// textbook online softmax in fp32 on the RAW scores (no deferred maximum, no pre-scaled Q: nothing is rounded to f16 before
// the exponential); P is split like every other operand.
#include <cstdlib>

#include "kernels.h"
#include "wca_common.h"

namespace wca {

namespace {

constexpr int KT = 64;
constexpr int TILE = 64 * 64;  // f16 elements of one 64-key x 64-dim tile
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ float xor16_maxf(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_maxf(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_sumf(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sumf(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Transposed LDS reads as inline asm with hand-counted lgkmcnt: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first
// ds_read_b64_tr_b16 BUILTIN of the loop (it cannot tell the transposed read from the LDS-DMA writes in flight), which drains the
// next tile's K/V prefetch on every tile (on this kernel the prefetch has landed by then anyway: 2.13 vs 2.20 ms per encoder layer
// at B = 64, within box variance -- kept because it removes the dependence on that timing). LDS operations return in issue order;
// the wait names its destination registers so that no consumer can be scheduled above it.
__device__ __forceinline__ unsigned lds_off_s(const void* p) { return (unsigned)(size_t)(const WCA_LDS char*)p; }
template <int OFF>
__device__ __forceinline__ half4 tr_read_asm(unsigned addr) {
  half4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ half8 b128_read_asm(unsigned addr) {
  half8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
  return r;
}
#define WCA_S_LGKM_WAIT4(N, A, B, C, D)                                                                       \
  do {                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(A), "+v"(B), "+v"(C), "+v"(D)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  } while (0)
#define WCA_S_LGKM_WAIT8(N, A, B, C, D, E, F, G, H)                                                                                  \
  do {                                                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(A), "+v"(B), "+v"(C), "+v"(D), "+v"(E), "+v"(F), "+v"(G), "+v"(H)::"memory");  \
    __builtin_amdgcn_sched_barrier(0);                                                                                               \
  } while (0)

// NW = waves per workgroup (32 query rows each): 4 everywhere (128-row blocks, two workgroups per CU = 2 waves per SIMD).
// PMC counters of this kernel on the encoder's 64 x 16 x 1500 x 1500 shape (profiles/r04_attn_split_pmc.txt): matrix pipe busy 47 %,
// vector ALU ~45 %, the two co-executing 7.5 % of the time, waves stalled at issue 49 %, parked at a wait 14.5 %, LDS index unit
// 16 %, no bank conflicts -- 0.95 PFLOP/s executed, the same plateau as the pair GEMMs. Two restructurings were built, verified
// against float64 and measured at that shape in round 4, neither pays:
//   * the wave's two 16-row sub-blocks software-pipelined against each other (S(1) under the exp2 / pair split of block 0, P.V(0) under
//     those of block 1: MFMAs under every vector stretch, K / V fragments of the whole tile held in registers, 212 VGPRs): 1.840 vs
//     1.853 ms -- removed;
//   * NW = 6 (192-row blocks, 384 threads, 166 VGPRs = three waves per SIMD, a third less K / V staging per query row): 2.116 vs
//     1.844 ms (the 8 staging pieces do not divide over 6 waves and the barrier waits for the two that stage twice) -- not
//     instantiated any more (the NW template parameter stays).
//   * one 8-wave workgroup per CU with its two wave groups in ENFORCED anti-phase (an extra barrier for waves 4-7, two barriers per tile,
//     ring of four slots: while one group is in its S^T MFMAs its SIMD partner is in exp2 / pair split): 1.963 vs 1.860 ms -- removed.
// What these null results say, and tools/micro/mfma_valu_coexec.hip confirms in isolation: on this chip dense vector work beside dense MFMAs
// is nearly ADDITIVE in time whichever wave it comes from (two waves per SIMD, 16 MFMAs + 96 v_fma per pair of iterations: 578 cycles
// against 256 of matrix pipe and 384 of vector issue), so the lever is the NUMBER of vector instructions, not where they sit.
// What did pay: the deferred running maximum (1.869 -> 1.837 ms) and the packed-f32 softmax below.
template <bool CAUSAL, bool CAPTURE, int NW = 4>
__global__ __launch_bounds__(NW * 64, (2 * NW * 64) / 256) void attn_split_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* lds = reinterpret_cast<half_t*>(smem);  // [slot][K hi | K lo | V hi | V lo]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int QB = NW * 32;  // query rows per workgroup
  const int n_qt = (a.nq + QB - 1) / QB;
  const int lid = xcd_remap(blockIdx.x, n_qt * a.H * a.B);
  const int bh = lid / n_qt;
  const int b = bh / a.H, h = bh - b * a.H;
  const int q_blk = (lid - bh * n_qt) * QB;
  const int q_wave = q_blk + wave * 32;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q = fr][dd = ks*32 + 8*fg + j], hi and lo halves
  half8 qh[2][2], ql[2][2];
  int qrow[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    qrow[sub] = q_wave + sub * 16 + fr;
    const int qc = qrow[sub] < a.nq ? qrow[sub] : a.nq - 1;
    const half_t* qp = a.Q + (long)b * a.q_bs + (long)qc * a.q_rs + h * 64 + fg * 8;
    qh[sub][0] = *reinterpret_cast<const half8*>(qp);
    qh[sub][1] = *reinterpret_cast<const half8*>(qp + 32);
    ql[sub][0] = *reinterpret_cast<const half8*>(qp + a.q_lo);
    ql[sub][1] = *reinterpret_cast<const half8*>(qp + a.q_lo + 32);
  }
  const float c_log2 = a.scale * LOG2E;
  // PRE (every variant that captures nothing): the Q pairs are multiplied by scale * log2(e) ONCE and split again (q c = hi' + lo' to
  // 2^-22), so the scores come out of the MFMAs in the log2 domain and the per-element multiply of the softmax disappears. The
  // capture variant keeps the raw q (the hooks see q.k * scale with one rounding).
  constexpr bool PRE = !CAPTURE;
  if (PRE) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const HalfPair pr = split_pair(((float)qh[sub][ks][j] + (float)ql[sub][ks][j]) * c_log2);
          qh[sub][ks][j] = pr.hi;
          ql[sub][ks][j] = pr.lo;
        }
  }
  const float cm = PRE ? 1.0f : c_log2;   // what turns a score difference into a log2 exponent

  int nk_eff = a.nk;
  if (CAUSAL) {
    const int qhi = q_blk + QB;
    nk_eff = qhi < a.nk ? qhi : a.nk;
  }
  const int nkt = (nk_eff + KT - 1) / KT;

  const half_t* Kb = a.K + (long)b * a.k_bs + h * 64;
  const half_t* Vb = a.V + (long)b * a.v_bs + h * 64;

  // a wave fills rows (wave*2 + i)*8 .. +7 of each of the four tiles: lane -> row + (lane >> 3), 16-byte chunk lane & 7 of the LDS
  // row, which holds global chunk (lane & 7) ^ swizzle(row) (K: swz128, conflict-free ds_read_b128; V: row & 6, transposed reads)
  auto stage = [&](int buf, int kt) {
    half_t* Kh = lds + buf * (4 * TILE);
    half_t* Kl = Kh + TILE;
    half_t* Vh = Kl + TILE;
    half_t* Vl = Vh + TILE;
    // the tile's 8 pieces of 8 key rows are dealt to the waves round-robin (NW = 6: waves 0, 1 stage two pieces, the others one)
#pragma unroll
    for (int i = 0; i < (8 + NW - 1) / NW; ++i) {
      const int piece = wave + NW * i;
      if (piece >= 8) break;   // (wave-uniform)
      const int rbase = piece * 8;
      const int r = rbase + (lane >> 3);
      int key = kt * KT + r;
      key = key < a.nk ? key : a.nk - 1;  // rows past nk: a duplicate of the last key, masked to -inf below
      const int ck = (lane & 7) ^ swz128(r);
      const int cv = (lane & 7) ^ (r & 6);
      const half_t* kp = Kb + (long)key * a.k_rs + ck * 8;
      const half_t* vp = Vb + (long)key * a.v_rs + cv * 8;
      glds16(kp, Kh + rbase * 64);
      glds16(kp + a.k_lo, Kl + rbase * 64);
      glds16(vp, Vh + rbase * 64);
      glds16(vp + a.v_lo, Vl + rbase * 64);
    }
  };

  f32x4 ot[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int d = 0; d < 4; ++d) ot[s][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane byte addresses of the V^T fragment reads in ring slot 0 (slot, hi / lo tile, k step and the second 16-key block are
  // immediates): lane (fr = 4 qd + pd, fg) reads key 4 fg + qd (+ 32 k2, + 16), dims 16 dt + 4 pd .. +3; chunk c = 2 dt + (pd >> 1)
  // sits at chunk position c ^ (key & 6), and key & 6 = 4 (fg & 1) | (qd & 2) for every one of this lane's keys
  unsigned vaddr[4];
  {
    const int qd = fr >> 2, pd = fr & 3;
    const int swz = (4 * (fg & 1)) | (qd & 2);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      vaddr[dt] = lds_off_s(lds) + 2u * (unsigned)((4 * fg + qd) * 64 + (((2 * dt + (pd >> 1)) ^ swz) << 3) + 4 * (pd & 1));
  }
  // per-lane byte addresses of the K fragment reads in ring slot 0, K hi tile (the K lo tile and the slot are added at the read):
  // fragment (ks, t) = row t*16 + fr, 16-byte chunk (4 ks + fg) ^ swz128(row)
  unsigned kaddr[2][4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = t * 16 + fr;
      kaddr[ks][t] = lds_off_s(lds) + 2u * (unsigned)(r * 64 + (((ks * 4 + fg) ^ swz128(r)) << 3));
    }
  float m_run[2] = {-INFINITY, -INFINITY};  // running row maximum of the scores (raw, or in the log2 domain when PRE)
  float l_part[2] = {0.f, 0.f};             // this lane's share of the row sum (its 16 keys per tile), reduced at the end

  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    wait_vm0();                      // this wave's requests of tile kt (the only ones in flight) have landed
    __builtin_amdgcn_s_barrier();    // ... and everyone's; every wave has also finished reading the other slot (tile kt - 1)
    asm volatile("" ::: "memory");
    if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
    // ---- S^T tile: st[sub][t][r] = S[q = fr (sub)][key = kt*64 + t*16 + 4*fg + r], three passes per 32-deep k step
    f32x4 st[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) st[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // all sixteen K fragments of the tile are requested at once (inline asm: hipcc otherwise issues each compiler-visible read two to six
      // MFMAs before its use behind an `s_waitcnt lgkmcnt(0)`, four exposed LDS latencies per tile); each group of MFMAs starts when its
      // four have landed (LDS operations return in issue order: kl ks0, kh ks0, kl ks1, kh ks1)
      constexpr int KLO = TILE * (int)sizeof(half_t);
      const unsigned sbk = (unsigned)(kt & 1) * (unsigned)(4 * TILE * sizeof(half_t));
      half8 kh[2][4], kl[2][4];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) kl[ks][t] = b128_read_asm<KLO>(kaddr[ks][t] + sbk);
#pragma unroll
        for (int t = 0; t < 4; ++t) kh[ks][t] = b128_read_asm<0>(kaddr[ks][t] + sbk);
      }
      // the small terms first, the hi.hi product last
      WCA_S_LGKM_WAIT4(12, kl[0][0], kl[0][1], kl[0][2], kl[0][3]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) st[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[0][t], qh[s][0], st[s][t], 0, 0, 0);
      WCA_S_LGKM_WAIT4(8, kh[0][0], kh[0][1], kh[0][2], kh[0][3]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) st[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[0][t], ql[s][0], st[s][t], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) st[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[0][t], qh[s][0], st[s][t], 0, 0, 0);
      WCA_S_LGKM_WAIT4(4, kl[1][0], kl[1][1], kl[1][2], kl[1][3]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) st[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[1][t], qh[s][1], st[s][t], 0, 0, 0);
      WCA_S_LGKM_WAIT4(0, kh[1][0], kh[1][1], kh[1][2], kh[1][3]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) st[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[1][t], ql[s][1], st[s][t], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) st[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[1][t], qh[s][1], st[s][t], 0, 0, 0);
    }

    if (CAPTURE) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (qrow[s] < a.nq) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int key4 = kt * KT + t * 16 + 4 * fg;
            if (key4 < a.cap_cols) {
              float* cp = a.cap + (long)b * a.cap_bs + (long)h * a.cap_hs + (long)qrow[s] * a.cap_ld + key4;
              *reinterpret_cast<f32x4*>(cp) = st[s][t] * a.scale;
            }
          }
        }
      }
    }
    const bool tail_tile = (kt * KT + KT > a.nk);
    const bool diag_tile = CAUSAL && (kt * KT + KT - 1 > q_wave);
    if (tail_tile || diag_tile) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * KT + t * 16 + 4 * fg + r;
            bool dead = key >= a.nk;
            if (CAUSAL) dead = dead || (key > qrow[s]);
            st[s][t][r] = dead ? -INFINITY : st[s][t][r];
          }
    }
    // The vector work of a tile costs wall time even beside MFMAs (tools/micro/mfma_valu_coexec.hip: with two waves per SIMD a stream of
    // 16 MFMAs + 96 plain vector instructions takes 0.9 x (pipe time + vector issue time)), so it is written on PAIRS of elements:
    // v_max3, v_pk_add_f32, v_cvt_pk_f16_f32 -- 4.5 instructions per score for difference, exp2, pair split and row sum instead of 9.
    float mx[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float m = fmaxf(fmaxf(st[s][0][0], st[s][0][1]), fmaxf(st[s][0][2], st[s][0][3]));
#pragma unroll
      for (int t = 1; t < 4; ++t) {
        m = fmaxf(fmaxf(m, st[s][t][0]), st[s][t][1]);   // v_max3_f32
        m = fmaxf(fmaxf(m, st[s][t][2]), st[s][t][3]);
      }
      mx[s] = xor32_maxf(xor16_maxf(m));
    }
    // DEFERRED running maximum (as in attention.hip): the reference point of a row is raised only when the tile maximum exceeds it
    // by more than 8 in the log2 domain (p <= 2^8: nowhere near the range of the f16 hi half, and p is carried as a pair anyway), or
    // when the first finite score of a row arrives. On random data SOME row of a wave grows its maximum in almost every tile, so the
    // undeferred branch (2 exp2 + 34 multiplies per lane) ran on nearly all of them.
    const float thr_raw = 8.0f / cm;
    if (__any((mx[0] > m_run[0] + thr_raw) || (mx[1] > m_run[1] + thr_raw) || (m_run[0] == -INFINITY && mx[0] != -INFINITY) ||
              (m_run[1] == -INFINITY && mx[1] != -INFINITY))) {  // wave-uniform
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float m_new = fmaxf(m_run[s], mx[s]);
        const float alpha = (m_run[s] == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f((m_run[s] - m_new) * cm);  // m_new finite here
        l_part[s] *= alpha;
#pragma unroll
        for (int d = 0; d < 4; ++d) ot[s][d] *= alpha;
        m_run[s] = m_new;
      }
    }
    // p = exp2((s - m) cm): the difference first (exact for nearby values); P^T fragments (B operand of O^T = V^T P^T): k-step k2
    // covers score tiles 2 k2, 2 k2 + 1, element j of the fragment = score (tile 2 k2 + (j >> 2), register j & 3)
    half8 ph[2][2], pl[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float mref = (m_run[s] == -INFINITY) ? 0.f : m_run[s];  // a row that has only seen masked keys: exp2(-inf) = 0
      const f32x2 mm = f32x2{mref, mref};
      f32x2 psum2 = f32x2{0.f, 0.f};
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        half2_ hh[4], ll[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // u -> (tile 2 k2 + (u >> 1), registers 2 (u & 1), 2 (u & 1) + 1)
          const int t = 2 * k2 + (u >> 1), r0 = 2 * (u & 1);
          f32x2 x = f32x2{st[s][t][r0], st[s][t][r0 + 1]} - mm;
          if (!PRE) x = x * f32x2{cm, cm};
          const f32x2 p = f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
          psum2 += p;
          hh[u] = __builtin_convertvector(p, half2_);   // v_cvt_pk_f16_f32 (RNE)
          // lo = f16(p - hi) by the mixed-precision fma (f32 p, f16 hi operand picked by op_sel; p - hi is exact before the one rounding):
          // one instruction per element, no conversion back to f32, no packing (hipcc does not form it from the plain expression)
          {
            const unsigned hu = __builtin_bit_cast(unsigned, hh[u]);
            unsigned lu;
            asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lu) : "v"(p[0]), "v"(hu));
            asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lu) : "v"(p[1]), "v"(hu));
            ll[u] = __builtin_bit_cast(half2_, lu);
          }
        }
        ph[s][k2] = half8{hh[0][0], hh[0][1], hh[1][0], hh[1][1], hh[2][0], hh[2][1], hh[3][0], hh[3][1]};
        pl[s][k2] = half8{ll[0][0], ll[0][1], ll[1][0], ll[1][1], ll[2][0], ll[2][1], ll[3][0], ll[3][1]};
      }
      l_part[s] += psum2[0] + psum2[1];
    }

    // ---- O^T += V^T P^T. V^T fragment (A operand): lane holds V[key(k)][d = dt*16 + fr], k order matching the P fragments:
    // j<4 -> key (2*k2)*16 + 4*fg + j, j>=4 -> key (2*k2+1)*16 + 4*fg + (j-4). The eight transposed reads of output block dt + 1
    // (hi / lo x two k steps x two 16-key blocks) are in flight under the twelve MFMAs of block dt.
    {
      const unsigned sb = (unsigned)(kt & 1) * (unsigned)(4 * TILE * sizeof(half_t));  // ring slot
      constexpr int VH = 2 * TILE * (int)sizeof(half_t), VL = 3 * TILE * (int)sizeof(half_t);  // V hi / V lo tile inside a slot
      constexpr int K2 = 32 * 64 * (int)sizeof(half_t), B16 = 16 * 64 * (int)sizeof(half_t);  // k step (32 keys), second 16-key block
      half4 c_[8], n_[8];
#define WCA_ISSUE_V(R, A)                                                                    \
  do {                                                                                       \
    R[0] = tr_read_asm<VH>(A);                                                               \
    R[1] = tr_read_asm<VH + B16>(A);                                                         \
    R[2] = tr_read_asm<VL>(A);                                                               \
    R[3] = tr_read_asm<VL + B16>(A);                                                         \
    R[4] = tr_read_asm<VH + K2>(A);                                                          \
    R[5] = tr_read_asm<VH + K2 + B16>(A);                                                    \
    R[6] = tr_read_asm<VL + K2>(A);                                                          \
    R[7] = tr_read_asm<VL + K2 + B16>(A);                                                    \
  } while (0)
      WCA_ISSUE_V(c_, vaddr[0] + sb);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        if (dt < 3) {
          WCA_ISSUE_V(n_, vaddr[dt + 1] + sb);
          WCA_S_LGKM_WAIT8(8, c_[0], c_[1], c_[2], c_[3], c_[4], c_[5], c_[6], c_[7]);
        } else {
          WCA_S_LGKM_WAIT8(0, c_[0], c_[1], c_[2], c_[3], c_[4], c_[5], c_[6], c_[7]);
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const half8 vh = __builtin_shufflevector(c_[4 * k2], c_[4 * k2 + 1], 0, 1, 2, 3, 4, 5, 6, 7);
          const half8 vl = __builtin_shufflevector(c_[4 * k2 + 2], c_[4 * k2 + 3], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            ot[s][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[s][k2], ot[s][dt], 0, 0, 0);
            ot[s][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[s][k2], ot[s][dt], 0, 0, 0);
            ot[s][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[s][k2], ot[s][dt], 0, 0, 0);
          }
        }
        if (dt < 3) {
#pragma unroll
          for (int i = 0; i < 8; ++i) c_[i] = n_[i];
        }
      }
#undef WCA_ISSUE_V
    }
  }

  // ---- epilogue: ot[s][dt][r] = O[q = fr][d = dt*16 + 4*fg + r]; a row's sum is spread over its four lane groups
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float l = xor32_sumf(xor16_sumf(l_part[s]));
    if (qrow[s] >= a.nq) continue;
    const float inv = 1.0f / l;
    half_t* op = a.O + (long)b * a.o_bs + (long)qrow[s] * a.o_rs + h * 64 + 4 * fg;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      half4 oh, ol;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const HalfPair pr = split_pair(ot[s][dt][r] * inv);
        oh[r] = pr.hi;
        ol[r] = pr.lo;
      }
      *reinterpret_cast<half4*>(op + dt * 16) = oh;
      *reinterpret_cast<half4*>(op + a.o_lo + dt * 16) = ol;
    }
  }
}



// ---------------------------------------------------------------------------------------------------------------------------------
// attn_split32_kernel (round 4): the encoder form (no mask, no capture) of the pair attention on v_mfma_f32_32x32x16_f16 -- attention.hip's
// attn32 structure carrying pairs. Why the other MFMA shape: a 16x16x32 MFMA occupies the matrix pipe for 16 cycles and the SIMD's vector
// issue for 8 of them, a 32x32x16 for 32 and 8 (MI355X_MICROARCH.md, issue-cost table) -- the same FLOPs block half as much vector issue,
// and this loop is issue-bound (profiles/r04_attn_split_pmc.txt: pipe time and vector time add up). What else the structure brings:
//   * S^T accumulators start at -m_running (a persistent C operand rewritten only when a maximum is raised): exp2 applies to the MFMA
//     result, no per-score subtraction;
//   * P^T reaches the P.V MFMAs with no lane movement (k order permuted, V^T fetched transposed in that order);
//   * K / V rows through buffer descriptors (rows past nk read as zero), per-lane offsets fixed, the tile advances through the scalar
//     offset: no vector instruction on DMA addresses; ring slots are compile-time (every LDS offset an immediate);
//   * per score: exp2, half a packed conversion, one mixed-precision fma for the lo half, half a packed add for the row sum.
// Measured at 64 x 16 x 1500 x 1500 (tools/split_bench.py, interleaved): 1.644-1.677 ms against 1.690-1.713 for the 16x16x32 kernel (-2...-3 %):
// the issue budget was not the whole story. What the experiments of this kernel say (all removed again): with K / V never re-staged
// (wrong results, timing only) 1.50 ms, so the DMA is 9 % of it; ONE workgroup per CU (one wave per SIMD) 1.95 ms, i.e. the second wave
// of a SIMD hides only a sixth of the first one's time; and a SOFTWARE-PIPELINED form (S^T of tile kt + 1 issued between the exp2 / pair
// split instructions of tile kt, K running one tile ahead of V in the same ring; 260 VGPRs = one wave per SIMD; verified against float64)
// also took 1.95 ms -- vector instructions interleaved with MFMAs inside ONE wave do not overlap them (tools/micro/mfma_valu_coexec.hip:
// 4 MFMA + 24 v_fma per iteration in one wave take 270 cycles against 128 of pipe), and two waves per SIMD need 2 x 256 registers.
// Same arithmetic contract as attn_split_kernel (three passes per product, fp32 online softmax with the deferred maximum, P and O split
// into pairs); the fp32 summation ORDER differs (k steps of 16 instead of 32, -m inside the accumulator), so results agree to fp32
// noise, not bit for bit -- tests/test_split_gpu.py compares both with float64.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float RESCALE_THR32 = 8.0f;

__device__ __forceinline__ float max3f_(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
template <int V>
struct IntC32 {
  static constexpr int value = V;
};
#define WCA_S_PIN8(A, B, C, D, E, F, G, H) asm volatile("" : "+v"(A), "+v"(B), "+v"(C), "+v"(D), "+v"(E), "+v"(F), "+v"(G), "+v"(H)::"memory")

// DROP (diagnostic, wca_test_set_attn_split_drop; the product runs DROP = 0): bit 0 leaves out K_lo Q_hi, bit 1 K_hi Q_lo, bit 2 V_lo P_hi,
// bit 3 V_hi P_lo -- the product-level ablation of the three-pass contract (tools/precision_ablation.py --attn-drop)
template <int DROP>
__global__ __launch_bounds__(256, 2) void attn_split32_kernel(AttnArgs a) {
  constexpr int NW = 4;  // waves per workgroup, 32 query rows each
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* lds = reinterpret_cast<half_t*>(smem);  // [slot][K hi | K lo | V hi | V lo]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  constexpr int QB = NW * 32;
  constexpr int TB = TILE * (int)sizeof(half_t);  // 8 KiB: one operand tile
  constexpr int SLOT_BYTES = 4 * TB;              // 32 KiB
  const int n_qt = (a.nq + QB - 1) / QB;
  const int lid = xcd_remap(blockIdx.x, n_qt * a.H * a.B);
  const int bh = lid / n_qt;
  const int b = bh / a.H, h = bh - b * a.H;
  const int q_wave = (lid - bh * n_qt) * QB + wave * 32;
  const int qrow = q_wave + l31;
  // (Round-5 experiment, removed: a STATIC issue priority for one of the two waves that share a SIMD -- they belong to two independent workgroups, run the same
  //  program, and PMC shows their matrix and vector phases co-executing only 10 % of the time -- chosen by wave slot (s_getreg HW_ID) or by workgroup parity:
  //  1.703 / 1.695 / 1.670 ms against 1.678-1.690 ms for the symmetric kernel at 64 x 16 x 1500 x 1500: nothing.)
  // Q fragments (B operand of S^T): lane holds Q[q = l31][dim = 16 ks + 8 hh + j]; the pair is multiplied by scale * log2(e) once and
  // split again, so the scores leave the MFMAs in the log2 domain
  const float c_log2 = a.scale * LOG2E;
  half8 qh[4], ql[4];
  {
    const int qc = qrow < a.nq ? qrow : a.nq - 1;
    const half_t* qp = a.Q + (long)b * a.q_bs + (long)qc * a.q_rs + h * 64 + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const half8 rh = *reinterpret_cast<const half8*>(qp + ks * 16);
      const half8 rl = *reinterpret_cast<const half8*>(qp + a.q_lo + ks * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const HalfPair pr = split_pair(((float)rh[j] + (float)rl[j]) * c_log2);
        qh[ks][j] = pr.hi;
        ql[ks][j] = pr.lo;
      }
    }
  }
  const int nkt = (a.nk + KT - 1) / KT;
  const int kbytes = (int)(((long)(a.nk - 1) * a.k_rs + 64) * 2), vbytes = (int)(((long)(a.nk - 1) * a.v_rs + 64) * 2);
  const half_t* Kb = a.K + (long)b * a.k_bs + h * 64;
  const half_t* Vb = a.V + (long)b * a.v_bs + h * 64;
  const __amdgpu_buffer_rsrc_t rkh = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Kb), 0, kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rkl = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Kb + a.k_lo), 0, kbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvh = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Vb), 0, vbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvl = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Vb + a.v_lo), 0, vbytes, 0x00020000);
  unsigned dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {  // 8 pieces of 8 key rows per tile and operand, two per wave
    const int r = (wave * 2 + i) * 8 + (lane >> 3);
    dk[i] = (unsigned)(r * a.k_rs * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4));
    dv[i] = (unsigned)(r * a.v_rs * 2 + (((lane & 7) ^ ((r & 2) << 1)) << 4));
  }
  const unsigned k_tile_bytes = (unsigned)(KT * a.k_rs * 2), v_tile_bytes = (unsigned)(KT * a.v_rs * 2);
  auto stage = [&](int buf, int kt) {
    half_t* Kh = lds + buf * (4 * TILE);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = (wave * 2 + i) * 8;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rkh, (WCA_LDS void*)(Kh + rbase * 64), 16, (int)dk[i], (int)(kt * k_tile_bytes), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rkl, (WCA_LDS void*)(Kh + TILE + rbase * 64), 16, (int)dk[i], (int)(kt * k_tile_bytes), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rvh, (WCA_LDS void*)(Kh + 2 * TILE + rbase * 64), 16, (int)dv[i], (int)(kt * v_tile_bytes), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rvl, (WCA_LDS void*)(Kh + 3 * TILE + rbase * 64), 16, (int)dv[i], (int)(kt * v_tile_bytes), 0, 0);
    }
  };

  // per-lane LDS byte addresses in slot 0 / the hi tiles (slot, lo tile and fragment index are immediates), as in attn32_kernel
  const int kswz = (l31 >> 1) & 7;
  const unsigned lds_base = lds_off_s(lds);
  unsigned ka[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ka[ks] = lds_base + (unsigned)(l31 * 128 + 16 * ((hh ^ kswz) & 1) + 32 * (ks ^ (kswz >> 1)));
  const int vq = (lane & 15) >> 2, vp = lane & 3, vg1 = (lane >> 4) & 1;
  const int vkey0 = 4 * hh + vq;
  const int vdim0 = 16 * vg1 + 4 * vp;
  const int vsw = (vkey0 & 2) << 1;
  unsigned va[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
    va[db] = lds_base + (unsigned)(2 * TB + vkey0 * 128 + (((4 * db + (vdim0 >> 3)) ^ vsw) << 4) + 2 * (vdim0 & 7));

  f32x16 ot[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
  float m_run = -INFINITY;  // running row maximum (log2 domain)
  float l_run = 0.f;        // this lane's share of the row sum (its 32 keys per tile); the two lane halves are combined at the end
  f32x16 cinit;             // -m_run in all 16 registers: the C operand of the first S^T MFMA of every key block
#pragma unroll
  for (int r = 0; r < 16; ++r) cinit[r] = 0.f;

  stage(0, 0);
  auto tile = [&](auto slot_c, int kt) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int SB = SLOT * SLOT_BYTES;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt (requested one tile ago) have landed
    __builtin_amdgcn_s_barrier();                      // ... everyone's; and every wave is done with the other slot
    asm volatile("" ::: "memory");
    if (kt + 1 < nkt) stage(SLOT ^ 1, kt + 1);   // (round 5: requested behind the S^T MFMAs instead 1.642 vs 1.648 ms, behind the first P.V block 1.671: nothing)

    // ---- S^T: st[kb][r] = s'(q = l31, key = 64 kt + 32 kb + (r&3) + 8 (r>>2) + 4 hh) - m_running; per k step the small terms first
    f32x16 st[2];
    {
      half8 kl0[4], kh0[4], kl1[4], kh1[4];
      kl0[0] = b128_read_asm<SB + TB>(ka[0]);
      kl0[1] = b128_read_asm<SB + TB>(ka[1]);
      kl0[2] = b128_read_asm<SB + TB>(ka[2]);
      kl0[3] = b128_read_asm<SB + TB>(ka[3]);
      kh0[0] = b128_read_asm<SB>(ka[0]);
      kh0[1] = b128_read_asm<SB>(ka[1]);
      kh0[2] = b128_read_asm<SB>(ka[2]);
      kh0[3] = b128_read_asm<SB>(ka[3]);
      kl1[0] = b128_read_asm<SB + TB + 32 * 128>(ka[0]);
      kl1[1] = b128_read_asm<SB + TB + 32 * 128>(ka[1]);
      kl1[2] = b128_read_asm<SB + TB + 32 * 128>(ka[2]);
      kl1[3] = b128_read_asm<SB + TB + 32 * 128>(ka[3]);
      kh1[0] = b128_read_asm<SB + 32 * 128>(ka[0]);
      kh1[1] = b128_read_asm<SB + 32 * 128>(ka[1]);
      kh1[2] = b128_read_asm<SB + 32 * 128>(ka[2]);
      kh1[3] = b128_read_asm<SB + 32 * 128>(ka[3]);
      if constexpr (DROP == 0) {
      WCA_S_LGKM_WAIT4(12, kl0[0], kl0[1], kl0[2], kl0[3]);
      st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl0[0], qh[0], cinit, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl0[ks], qh[ks], st[0], 0, 0, 0);
      WCA_S_LGKM_WAIT4(8, kh0[0], kh0[1], kh0[2], kh0[3]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0[ks], ql[ks], st[0], 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0[ks], qh[ks], st[0], 0, 0, 0);
      WCA_S_LGKM_WAIT4(4, kl1[0], kl1[1], kl1[2], kl1[3]);
      st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl1[0], qh[0], cinit, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl1[ks], qh[ks], st[1], 0, 0, 0);
      WCA_S_LGKM_WAIT4(0, kh1[0], kh1[1], kh1[2], kh1[3]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1[ks], ql[ks], st[1], 0, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1[ks], qh[ks], st[1], 0, 0, 0);
      } else {   // ablation forms: the same order with the dropped passes left out (the large term last)
        WCA_S_LGKM_WAIT4(0, kh1[0], kh1[1], kh1[2], kh1[3]);
        WCA_S_PIN8(kl0[0], kl0[1], kl0[2], kl0[3], kh0[0], kh0[1], kh0[2], kh0[3]);
        WCA_S_PIN8(kl1[0], kl1[1], kl1[2], kl1[3], kh1[0], kh1[1], kh1[2], kh1[3]);
        st[0] = cinit;
        st[1] = cinit;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (!(DROP & 1)) st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl0[ks], qh[ks], st[0], 0, 0, 0);
          if (!(DROP & 1)) st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl1[ks], qh[ks], st[1], 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (!(DROP & 2)) st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0[ks], ql[ks], st[0], 0, 0, 0);
          if (!(DROP & 2)) st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1[ks], ql[ks], st[1], 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0[ks], qh[ks], st[0], 0, 0, 0);
          st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1[ks], qh[ks], st[1], 0, 0, 0);
        }
      }
    }
    // V^T fragments (hi and lo) of the first 32 output dims: requested now, they land under the softmax
    half4 v0ha[4], v0hb[4], v0la[4], v0lb[4], v1ha[4], v1hb[4], v1la[4], v1lb[4];
#define WCA_ISSUE_V32(P, DB)                                           \
  do {                                                                 \
    P##ha[0] = tr_read_asm<SB + 0 * 2048>(va[DB]);                     \
    P##hb[0] = tr_read_asm<SB + 0 * 2048 + 1024>(va[DB]);              \
    P##ha[1] = tr_read_asm<SB + 1 * 2048>(va[DB]);                     \
    P##hb[1] = tr_read_asm<SB + 1 * 2048 + 1024>(va[DB]);              \
    P##ha[2] = tr_read_asm<SB + 2 * 2048>(va[DB]);                     \
    P##hb[2] = tr_read_asm<SB + 2 * 2048 + 1024>(va[DB]);              \
    P##ha[3] = tr_read_asm<SB + 3 * 2048>(va[DB]);                     \
    P##hb[3] = tr_read_asm<SB + 3 * 2048 + 1024>(va[DB]);              \
    P##la[0] = tr_read_asm<SB + TB + 0 * 2048>(va[DB]);                \
    P##lb[0] = tr_read_asm<SB + TB + 0 * 2048 + 1024>(va[DB]);         \
    P##la[1] = tr_read_asm<SB + TB + 1 * 2048>(va[DB]);                \
    P##lb[1] = tr_read_asm<SB + TB + 1 * 2048 + 1024>(va[DB]);         \
    P##la[2] = tr_read_asm<SB + TB + 2 * 2048>(va[DB]);                \
    P##lb[2] = tr_read_asm<SB + TB + 2 * 2048 + 1024>(va[DB]);         \
    P##la[3] = tr_read_asm<SB + TB + 3 * 2048>(va[DB]);                \
    P##lb[3] = tr_read_asm<SB + TB + 3 * 2048 + 1024>(va[DB]);         \
  } while (0)
    WCA_ISSUE_V32(v0, 0);
    if (kt * KT + KT > a.nk) {  // keys past nk (last tile only): wave-uniform
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          st[kb][r] = key >= a.nk ? -INFINITY : st[kb][r];
        }
    }
    float mx;
    {
      float m0 = max3f_(st[0][0], st[0][1], st[0][2]), m1 = max3f_(st[0][3], st[0][4], st[0][5]);
      float m2 = max3f_(st[0][6], st[0][7], st[0][8]), m3 = max3f_(st[0][9], st[0][10], st[0][11]);
      m0 = max3f_(m0, st[0][12], st[0][13]);
      m1 = max3f_(m1, st[0][14], st[0][15]);
      m2 = max3f_(m2, st[1][0], st[1][1]);
      m3 = max3f_(m3, st[1][2], st[1][3]);
      m0 = max3f_(m0, st[1][4], st[1][5]);
      m1 = max3f_(m1, st[1][6], st[1][7]);
      m2 = max3f_(m2, st[1][8], st[1][9]);
      m3 = max3f_(m3, st[1][10], st[1][11]);
      m0 = max3f_(m0, st[1][12], st[1][13]);
      m1 = max3f_(m1, st[1][14], st[1][15]);
      mx = fmaxf(max3f_(m0, m1, m2), m3);
      mx = xor32_maxf(mx);
    }
    // deferred maximum: raised when the tile maximum of (s' - m) exceeds the threshold, or anything finite arrives while m is still -inf
    if (__any((mx > RESCALE_THR32) || (m_run == -INFINITY && mx != -INFINITY))) {
      const float m_eff = (m_run == -INFINITY) ? 0.f : m_run;
      const float m_new = fmaxf(m_run, mx + m_eff);
      const float delta = (m_new == -INFINITY) ? 0.f : m_new - m_eff;
      const float alpha = (m_run == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] -= delta;
      m_run = m_new;
      const float c0 = (m_new != -INFINITY) ? -m_new : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) cinit[r] = c0;
    }
    // p = exp2(s' - m) as a pair; P^T fragment of k step s4 (16 keys): registers 8 (s4 & 1) .. +7 of key block s4 >> 1
    half8 ph[4], pl[4];
    {
      f32x2 psum = f32x2{0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        half2_ hq[4], lq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r0 = 8 * (s4 & 1) + 2 * u;
          const f32x2 p = f32x2{__builtin_amdgcn_exp2f(st[s4 >> 1][r0]), __builtin_amdgcn_exp2f(st[s4 >> 1][r0 + 1])};
          psum += p;
          hq[u] = __builtin_convertvector(p, half2_);
          const unsigned hu = __builtin_bit_cast(unsigned, hq[u]);
          unsigned lu;
          asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lu) : "v"(p[0]), "v"(hu));
          asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lu) : "v"(p[1]), "v"(hu));
          lq[u] = __builtin_bit_cast(half2_, lu);
        }
        ph[s4] = half8{hq[0][0], hq[0][1], hq[1][0], hq[1][1], hq[2][0], hq[2][1], hq[3][0], hq[3][1]};
        pl[s4] = half8{lq[0][0], lq[0][1], lq[1][0], lq[1][1], lq[2][0], lq[2][1], lq[3][0], lq[3][1]};
      }
      l_run += psum[0] + psum[1];
    }
    // ---- O^T += V^T P^T: A operand element j of lane half hh = V[key 16 s4 + 8 (j>>2) + 4 hh + (j&3)][d = 32 db + l31]
    WCA_S_LGKM_WAIT8(0, v0ha[0], v0hb[0], v0ha[1], v0hb[1], v0ha[2], v0hb[2], v0ha[3], v0hb[3]);
    WCA_S_PIN8(v0la[0], v0lb[0], v0la[1], v0lb[1], v0la[2], v0lb[2], v0la[3], v0lb[3]);
    WCA_ISSUE_V32(v1, 1);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const half8 wh = __builtin_shufflevector(v0ha[s4], v0hb[s4], 0, 1, 2, 3, 4, 5, 6, 7);
      const half8 wl = __builtin_shufflevector(v0la[s4], v0lb[s4], 0, 1, 2, 3, 4, 5, 6, 7);
      if (!(DROP & 4)) ot[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ph[s4], ot[0], 0, 0, 0);
      if (!(DROP & 8)) ot[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, pl[s4], ot[0], 0, 0, 0);
      ot[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ph[s4], ot[0], 0, 0, 0);
    }
    WCA_S_LGKM_WAIT8(0, v1ha[0], v1hb[0], v1ha[1], v1hb[1], v1ha[2], v1hb[2], v1ha[3], v1hb[3]);
    WCA_S_PIN8(v1la[0], v1lb[0], v1la[1], v1lb[1], v1la[2], v1lb[2], v1la[3], v1lb[3]);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const half8 wh = __builtin_shufflevector(v1ha[s4], v1hb[s4], 0, 1, 2, 3, 4, 5, 6, 7);
      const half8 wl = __builtin_shufflevector(v1la[s4], v1lb[s4], 0, 1, 2, 3, 4, 5, 6, 7);
      if (!(DROP & 4)) ot[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ph[s4], ot[1], 0, 0, 0);
      if (!(DROP & 8)) ot[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, pl[s4], ot[1], 0, 0, 0);
      ot[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ph[s4], ot[1], 0, 0, 0);
    }
#undef WCA_ISSUE_V32
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    tile(IntC32<0>{}, kt);
    if (kt + 1 < nkt) tile(IntC32<1>{}, kt + 1);
  }

  // ---- epilogue: ot[db][r] = O[q = l31][d = 32 db + (r&3) + 8 (r>>2) + 4 hh]; the two halves of a row are swapped pairwise so that
  // each lane stores 16 contiguous bytes of the hi row and of the lo row
  const float inv = 1.0f / xor32_sumf(l_run);
  half_t* op = a.O + (long)b * a.o_bs + (long)qrow * a.o_rs + h * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      unsigned wh[2][2], wl[2][2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const Half2Pair pr = split_pair2(ot[db][4 * (2 * g2 + e) + 2 * k2] * inv, ot[db][4 * (2 * g2 + e) + 2 * k2 + 1] * inv);
          wh[e][k2] = __builtin_bit_cast(unsigned, pr.hi);
          wl[e][k2] = __builtin_bit_cast(unsigned, pr.lo);
        }
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      u32x4 oh, ol;
      {
        auto r0 = __builtin_amdgcn_permlane32_swap(wh[0][0], wh[1][0], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(wh[0][1], wh[1][1], false, false);
        oh = u32x4{r0[0], r1[0], r0[1], r1[1]};
        auto s0 = __builtin_amdgcn_permlane32_swap(wl[0][0], wl[1][0], false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(wl[0][1], wl[1][1], false, false);
        ol = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
      if (qrow < a.nq) {
        *reinterpret_cast<u32x4*>(op + 32 * db + 16 * g2 + 8 * hh) = oh;
        *reinterpret_cast<u32x4*>(op + a.o_lo + 32 * db + 16 * g2 + 8 * hh) = ol;
      }
    }
}

}  // namespace

hipError_t launch_attention_split(const AttnArgs& a, hipStream_t s) {
  if (a.nq <= 0 || a.B <= 0) return hipSuccess;
  if (a.nk <= 0) return hipErrorInvalidValue;
  if ((a.q_rs % 8) || (a.k_rs % 8) || (a.v_rs % 8) || (a.o_rs % 4)) return hipErrorInvalidValue;
  if ((a.q_lo % 8) || (a.k_lo % 8) || (a.v_lo % 8) || (a.o_lo % 4) || a.q_lo <= 0 || a.k_lo <= 0 || a.v_lo <= 0 || a.o_lo <= 0) return hipErrorInvalidValue;
  if (a.cap != nullptr && ((a.cap_ld % 4) != 0 || a.cap_ld < ((a.cap_cols + 3) & ~3))) return hipErrorInvalidValue;
  const bool cap = a.cap != nullptr && a.cap_cols > 0;
  const size_t shmem = 2 * 4 * TILE * sizeof(half_t);  // 64 KiB: two slots of K hi | K lo | V hi | V lo
#define WCA_LAUNCH_AS(C, P, W)                                                                                               \
  do {                                                                                                                       \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_split_kernel<C, P, W>),                            \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                             \
    if (e != hipSuccess) return e;                                                                                           \
    hipLaunchKernelGGL((attn_split_kernel<C, P, W>), dim3(((a.nq + (W) * 32 - 1) / ((W) * 32)) * a.H * a.B), dim3((W) * 64), shmem, s, a); \
  } while (0)
  // the encoder form (no mask, no capture) on the 32x32x16 kernel; AttnArgs.variant 1 (tests / A-B: WCA_ATTN_SPLIT_VARIANT=1) keeps the 16x16x32 one
  const int variant = a.variant ? a.variant : debug_switch(DBG_ATTN_SPLIT_VARIANT);
  if (!a.causal && !cap && a.nq >= 64 && variant != 1 && (a.o_rs % 8) == 0 && (a.o_lo % 8) == 0) {
#define WCA_LAUNCH_A32(D)                                                                                                             \
  do {                                                                                                                                \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_split32_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    if (e != hipSuccess) return e;                                                                                                    \
    hipLaunchKernelGGL(attn_split32_kernel<D>, dim3(((a.nq + 127) / 128) * a.H * a.B), dim3(256), shmem, s, a);                       \
  } while (0)
    switch (debug_switch(DBG_ATTN_SPLIT_DROP)) {   // 0 in the product; the other forms are the ablation's (wca_test_set_attn_split_drop)
      case 0: WCA_LAUNCH_A32(0); break;
      case 1: WCA_LAUNCH_A32(1); break;
      case 2: WCA_LAUNCH_A32(2); break;
      case 3: WCA_LAUNCH_A32(3); break;
      case 4: WCA_LAUNCH_A32(4); break;
      case 8: WCA_LAUNCH_A32(8); break;
      case 9: WCA_LAUNCH_A32(9); break;
      case 12: WCA_LAUNCH_A32(12); break;
      case 15: WCA_LAUNCH_A32(15); break;
      default: return hipErrorInvalidValue;
    }
#undef WCA_LAUNCH_A32
    return hipGetLastError();
  }
  if (a.causal) {
    if (cap) WCA_LAUNCH_AS(true, true, 4); else WCA_LAUNCH_AS(true, false, 4);
  } else {
    if (cap) WCA_LAUNCH_AS(false, true, 4); else WCA_LAUNCH_AS(false, false, 4);
  }
#undef WCA_LAUNCH_AS
  return hipGetLastError();
}

}  // namespace wca
