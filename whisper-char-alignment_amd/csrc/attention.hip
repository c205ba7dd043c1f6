// Multi-head attention for gfx950 with head_dim 64 (all Whisper sizes), one kernel for
//   * encoder self-attention            (reference: timing.py:58 -> whisper AudioEncoder blocks)
//   * decoder causal self-attention
//   * decoder cross-attention with CAPTURE of the pre-softmax logits qk = (q*s)(k*s)^T, which is
//     exactly what the reference's forward hooks collect (timing.py:50-55, outs[-1]).
//
// Structure: 256-thread workgroup = 4 waves, each wave owns 32 query rows (two 16-row subtiles);
// 64-key K/V tiles are streamed HBM -> LDS with LDS-DMA (ring of 3, two tiles ahead). Scores are computed
// TRANSPOSED (S^T = K Q^T, v_mfma_f32_16x16x32_f16) so a query row lives on one lane column and
// the online-softmax statistics are lane-local (+2 cross-lane steps). P^T feeds the second MFMA
// (O^T = V^T P^T) straight from registers; V^T fragments come from ds_read_b64_tr_b16.
#include <cstdlib>

#include "kernels.h"
#include "wca_common.h"

namespace wca {

namespace {

constexpr int KT = 64;               // keys per tile
constexpr int TILE = 64 * 64;        // f16 elements of one K or V tile
constexpr float LOG2E = 1.4426950408889634f;
// Deferred running-max update of the lazy online softmax: a row's running maximum m is only raised (and O, l rescaled)
// when a tile's scores exceed it by more than RESCALE_THR in the log2 domain; until then p = exp2(s' - m) may be as large as
// 2^THR = 256 -- exact to f16's 11 bits like any other p (f16 keeps its relative precision up to 65504), sums are fp32.
// Without the threshold the wave-uniform rescale branch fires on almost every tile (32 rows per wave: on random data at
// least one row's maximum grows in 75-100 % of the tiles), ~90 extra vector instructions per wave-tile.
constexpr float RESCALE_THR = 8.0f;

// max over the lanes {l, l^16} / {l, l^32} without LDS: the swap returns {own, partner} in some order
__device__ __forceinline__ float xor16_max(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ half4 tr_read4(const half_t* p) {
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WCA_LDS s16x4*)(p));
  return __builtin_bit_cast(half4, r);
}

// ---- LDS reads the compiler must not schedule or wait for itself. hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the
// first ds_read_b64_tr_b16 *builtin* of the loop -- it cannot tell the transposed read from the LDS-DMA writes in flight --
// which drains the K/V prefetch on every tile; and it issues a compiler-visible ds_read_b128 only right before its MFMA
// (one read in flight, `lgkmcnt(0)` each). These inline-asm forms are invisible to that bookkeeping: the caller counts
// lgkmcnt itself (LDS operations return in issue order) and names the destinations in the wait statement, so that no
// consumer can be scheduled above the wait (cdna_hip_programming.md 5.7, form (ii)).
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(const WCA_LDS char*)p; }
template <int OFF>
__device__ __forceinline__ half8 lds_read_b128_asm(unsigned addr) {
  half8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ half4 lds_read_tr_asm(unsigned addr) {
  half4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
  return r;
}
#define WCA_LGKM_WAIT4(N, A, B, C, D)                                                                         \
  do {                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(A), "+v"(B), "+v"(C), "+v"(D)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  } while (0)
#define WCA_LGKM_WAIT8(N, A, B, C, D, E, F, G, H)                                                             \
  do {                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(A), "+v"(B), "+v"(C), "+v"(D), "+v"(E), "+v"(F), "+v"(G), "+v"(H)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  } while (0)

template <bool CAUSAL, bool CAPTURE, bool STAMP = false>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* lds = reinterpret_cast<half_t*>(smem);  // [buf][K tile | V tile]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  // 1-D grid, XCD-aware: logical id = ((b*H + h) * n_qt + qt); the bijective remap gives every XCD a contiguous
  // range of logical ids, so all query tiles of one (batch, head) run on ONE XCD and share its K/V in that L2.
  const int n_qt = (a.nq + 127) / 128;
  const int lid = xcd_remap(blockIdx.x, n_qt * a.H * a.B);
  const int bh = lid / n_qt;
  const int b = bh / a.H, h = bh - b * a.H;
  const int q_blk = (lid - bh * n_qt) * 128;
  const int q_wave = q_blk + wave * 32;

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q = fr][dd = ks*32 + 8*fg + j]
  half8 qf[2][2];
  int qrow[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    qrow[sub] = q_wave + sub * 16 + fr;
    const int qc = qrow[sub] < a.nq ? qrow[sub] : a.nq - 1;
    const half_t* qp = a.Q + (long)b * a.q_bs + (long)qc * a.q_rs + h * 64 + fg * 8;
    qf[sub][0] = *reinterpret_cast<const half8*>(qp);
    qf[sub][1] = *reinterpret_cast<const half8*>(qp + 32);
  }

  // LAZY (every variant without capture): the softmax runs on s' = (q * scale*log2e) . k  -  m_running, produced
  // DIRECTLY by the S^T MFMAs: Q fragments are pre-multiplied once (f16 round of q*c; the capture variant keeps the
  // exact fp32 scaling of the raw logits the reference's hooks see) and the accumulator starts at -m_running, so the
  // per-element fma(s, c, -m*c) of the textbook form disappears from the VALU-bound loop (32 of ~150 VALU
  // instructions per 64-key tile). A row whose maximum grows is fixed up in the (rare, wave-uniform) rescale branch.
  constexpr bool LAZY = !CAPTURE;
  const float c_log2 = a.scale * LOG2E;
  if (LAZY) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[sub][ks][j] = (half_t)((float)qf[sub][ks][j] * c_log2);
  }

  int nk_eff = a.nk;
  if (CAUSAL) {
    const int qhi = q_blk + 128;
    nk_eff = qhi < a.nk ? qhi : a.nk;
  }
  const int nkt = (nk_eff + KT - 1) / KT;

  const half_t* Kb = a.K + (long)b * a.k_bs + h * 64;
  const half_t* Vb = a.V + (long)b * a.v_bs + h * 64;

  auto stage = [&](int buf, int kt) {
    half_t* Kt = lds + buf * (2 * TILE);
    half_t* Vt = Kt + TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = (wave * 2 + i) * 8;
      const int r = rbase + (lane >> 3);
      int key = kt * KT + r;
      key = key < a.nk ? key : a.nk - 1;
      const int ck = (lane & 7) ^ swz128(r);
      const int cv = (lane & 7) ^ (r & 6);
      glds16(Kb + (long)key * a.k_rs + ck * 8, Kt + rbase * 64);
      glds16(Vb + (long)key * a.v_rs + cv * 8, Vt + rbase * 64);
    }
  };

  f32x4 ot[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int d = 0; d < 4; ++d) ot[s][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.0f;
  float m_run[2] = {-INFINITY, -INFINITY};
  float l_run[2] = {0.f, 0.f};

  // K/V ring of 3 tiles, LDS-DMA two tiles ahead, counted vmcnt, ONE raw barrier per key tile: at the top of
  // tile kt every wave has waited for its own requests of tile kt (the 4 of tile kt+1 may stay in flight), so
  // the barrier publishes tile kt (RAW) and also proves that everyone is done with tile kt-1, whose slot is
  // refilled with tile kt+2 right after it (WAR).
  stage(0, 0);
  if (nkt > 1) stage(1, 1);
  int slot = 0;
#define WCA_STAMP(IDX)                                                                      \
  do {                                                                                     \
    if (STAMP) {                                                                           \
      unsigned long long t_;                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      if (lane == 0 && blockIdx.x < 4 && kt < 32) a.dbg[((blockIdx.x * 4 + wave) * 32 + kt) * 8 + (IDX)] = t_; \
    }                                                                                      \
  } while (0)
  for (int kt = 0; kt < nkt; ++kt) {
    WCA_STAMP(0);
    if (kt + 1 < nkt) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    WCA_STAMP(1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WCA_STAMP(2);
    if (kt + 2 < nkt) {
      int nslot = slot + 2;
      nslot = nslot >= 3 ? nslot - 3 : nslot;
      stage(nslot, kt + 2);
    }
    const half_t* Kt = lds + slot * (2 * TILE);
    const half_t* Vt = Kt + TILE;
    slot = (slot == 2) ? 0 : slot + 1;

    // ---- S^T tile: st[sub][t][r] = S[q = fr (sub)][key = kt*64 + t*16 + 4*fg + r]
    f32x4 st[2][4];
    {
      half8 kf0[4], kf1[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = t * 16 + fr;
        kf0[t] = *reinterpret_cast<const half8*>(Kt + r * 64 + (((0 + fg) ^ swz128(r)) << 3));
        kf1[t] = *reinterpret_cast<const half8*>(Kt + r * 64 + (((4 + fg) ^ swz128(r)) << 3));
      }
      f32x4 cinit[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float c0 = (LAZY && m_run[s] != -INFINITY) ? -m_run[s] : 0.f;  // LAZY: m_run is kept in the scaled log2 domain
        cinit[s] = f32x4{c0, c0, c0, c0};
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) st[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0[t], qf[s][0], cinit[s], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) st[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1[t], qf[s][1], st[s][t], 0, 0, 0);
    }
    WCA_STAMP(3);

    // ---- online softmax on the RAW scores (scale > 0 commutes with max); p = exp2(s*c - m*c), c = scale*log2(e).
    // Both 16-row subtiles advance together (one combined rescale branch) so their dependency chains interleave.
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
    // masks are only materialised on tiles that need them (last key tile / causal diagonal): wave-uniform branch
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
    float mx[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float m = st[s][0][0];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; r += 2) m = fmaxf(fmaxf(m, st[s][t][r]), st[s][t][r + 1]);
      mx[s] = m;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) mx[s] = xor16_max(mx[s]);
#pragma unroll
    for (int s = 0; s < 2; ++s) mx[s] = xor32_max(mx[s]);
    if (LAZY) {
      // st holds s' - m_eff with m_eff = m_run (0 while m_run is still -inf). The maximum is raised when the tile maximum of
      // that exceeds RESCALE_THR (or anything finite arrives while m_run is -inf).
      if (__any((mx[0] > RESCALE_THR) || (mx[1] > RESCALE_THR) || (m_run[0] == -INFINITY && mx[0] != -INFINITY) ||
                (m_run[1] == -INFINITY && mx[1] != -INFINITY))) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const float m_eff = (m_run[s] == -INFINITY) ? 0.f : m_run[s];
          const float m_new = fmaxf(m_run[s], mx[s] + m_eff);
          const float delta = (m_new == -INFINITY) ? 0.f : m_new - m_eff;  // still to be subtracted from this tile's scores
          const float alpha = (m_run[s] == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run[s] - m_new);
          l_run[s] *= alpha;
#pragma unroll
          for (int d = 0; d < 4; ++d) ot[s][d] *= alpha;
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[s][t][r] -= delta;
          m_run[s] = m_new;
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[s][t][r] = __builtin_amdgcn_exp2f(st[s][t][r]);
    } else {
    if (__any((mx[0] > m_run[0]) || (mx[1] > m_run[1]))) {  // some row's running max grows: textbook rescale
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float m_new = fmaxf(m_run[s], mx[s]);
        const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f((m_run[s] - m_new) * c_log2);
        l_run[s] *= alpha;
#pragma unroll
        for (int d = 0; d < 4; ++d) ot[s][d] *= alpha;
        m_run[s] = m_new;
      }
    }
    float mc[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) mc[s] = (m_run[s] == -INFINITY) ? 0.f : m_run[s] * c_log2;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[s][t][r] = __builtin_amdgcn_exp2f(fmaf(st[s][t][r], c_log2, -mc[s]));
    }
    // P^T fragments (B operand of O^T = V^T P^T): k-step k2 covers score tiles 2*k2, 2*k2+1
    half8 pf[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        half8 f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f[r] = (half_t)st[s][2 * k2][r];
          f[4 + r] = (half_t)st[s][2 * k2 + 1][r];
        }
        pf[s][k2] = f;
      }
    // row sums on the matrix pipe: D[i][q] = sum_k 1 * P^T[k][q] for every i, i.e. the sum over the tile's 64
    // keys of exactly the f16-rounded probabilities that enter P.V (no VALU adds, no cross-lane step)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 rsum = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[s][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      rsum = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[s][1], rsum, 0, 0, 0);
      l_run[s] += rsum[0];
    }
    WCA_STAMP(4);

    // ---- O^T += V^T P^T. V^T fragment (A operand): lane holds V[key(k)][d = dt*16 + fr],
    // k order matches pf: j<4 -> key (2*k2)*16 + 4*fg + j, j>=4 -> key (2*k2+1)*16 + 4*fg + (j-4).
    {
      const int qd = fr >> 2, pd = fr & 3;  // this lane supplies row qd, columns 4*pd..4*pd+3 of its group's block
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = dt * 16 + 4 * pd;
        const int c = d >> 3, w = d & 7;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int key0 = (2 * k2) * 16 + 4 * fg + qd;
          const int key1 = key0 + 16;
          const half4 v0 = tr_read4(Vt + key0 * 64 + ((c ^ (key0 & 6)) << 3) + w);
          const half4 v1 = tr_read4(Vt + key1 * 64 + ((c ^ (key1 & 6)) << 3) + w);
          half8 vf;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            vf[r] = v0[r];
            vf[4 + r] = v1[r];
          }
#pragma unroll
          for (int s = 0; s < 2; ++s)
            ot[s][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[s][k2], ot[s][dt], 0, 0, 0);
        }
      }
    }
    WCA_STAMP(5);
  }
#undef WCA_STAMP

  // ---- epilogue: ot[s][dt][r] = O[q = fr][d = dt*16 + 4*fg + r]
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (qrow[s] >= a.nq) continue;
    const float inv = 1.0f / l_run[s];
    half_t* op = a.O + (long)b * a.o_bs + (long)qrow[s] * a.o_rs + h * 64 + 4 * fg;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      half4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (half_t)(ot[s][dt][r] * inv);
      *reinterpret_cast<half4*>(op + dt * 16) = o;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Encoder self-attention (no mask, no capture; 589.8 GFLOP per layer at batch 64): the same flash structure on
// v_mfma_f32_32x32x16_f16. Why a second kernel: the loop above is bound by VALU ISSUE, not by the matrix pipe
// (per 64-key tile and wave 36 MFMAs = 576 pipe cycles, but ~150 vector instructions = ~740 issue cycles, and every
// 16x16x32 MFMA holds the SIMD's vector issue for 8 of its 16 cycles: 3 waves per SIMD measured 3420 cycles per three
// wave-tiles against 3 x (736 + 36 x 8) = 3072 of vector issue). A 32x32x16 MFMA does twice the work per 8 blocked issue
// cycles, and in its accumulator layout a query row is ONE lane column with 16 keys per 32-key block in registers, so
//   * S^T = K Q^T: 8 MFMAs per wave-tile (2 key blocks x 4 k-steps), accumulator initialised to -m_running (Q carries
//     scale * log2 e), i.e. exp2 applies directly to the MFMA result;
//   * row max: v_max3 over the lane's 32 values + one v_permlane32_swap (the other 32 keys of the row live in lane ^ 32);
//   * P^T needs NO lane movement to become the B operand of O^T = V^T P^T: registers 8s .. 8s+7 of a key block are the
//     fragment of k-step s, with the k order permuted (element j of lane half h = key 16s + 8(j>>2) + 4h + (j&3)); the
//     V^T fragments are fetched with ds_read_b64_tr_b16 in exactly that key order;
//   * row sums on the matrix pipe (ones x P^T), 4 MFMAs;
// per wave-tile: 20 MFMAs (640 pipe cycles, 160 blocked issue cycles) and ~32 v_exp + ~60 other vector instructions.
// Same staging as above: 256-thread workgroup, 32 query rows per wave, ring of three 64-key K/V tiles filled by LDS-DMA two
// tiles ahead, counted vmcnt, one raw barrier per tile. K image: chunk ^ ((key >> 1) & 7) (conflict-free ds_read_b128 for
// the 32-row A operand); V image: chunk ^ ((key & 2) << 1) (conflict-free transposed reads).
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <int V>
struct IntC {
  static constexpr int value = V;
};

__device__ __forceinline__ float xor32_sumf(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// VSUM (experiment, AttnArgs.variant 3; NOT the default): row sums as fp32 VALU adds of the lane's 32 probabilities per tile (per-lane
// partial, the two lane halves of a row combined once at the end) instead of four `ones x P^T` MFMAs per tile -- a fifth of the tile's
// matrix work. Interleaved A/B at 64 x 16 x 1500 x 1500 (tools/attn_ab.py 64 10 2,3): 0.708 vs 0.724 ms median on one box, 0.701 vs 0.699 on
// another. Rejected for its numerics: the MFMA form sums exactly the f16-rounded probabilities that enter P.V, so the rounding of P cancels
// between numerator and denominator (a row dominated by one key returns that V row exactly); summing the unrounded fp32 values adds
// ~2^-12 relative noise per layer, and the bench's 301-utterance f16 parity leg went from 58 to 139 boundaries outside one frame with it.
template <bool STAMP, bool VSUM = false>
__global__ __launch_bounds__(256) void attn32_kernel(AttnArgs a) {
  constexpr int NW = 4;  // waves per workgroup, 32 query rows each
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* lds = reinterpret_cast<half_t*>(smem);  // [slot][K tile | V tile]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  constexpr int QB = NW * 32;  // query rows per workgroup
  constexpr int SLOT_BYTES = 2 * TILE * (int)sizeof(half_t);  // 16 KiB: K tile | V tile
  const int n_qt = (a.nq + QB - 1) / QB;
  const int lid = xcd_remap(blockIdx.x, n_qt * a.H * a.B);
  const int bh = lid / n_qt;
  const int b = bh / a.H, h = bh - b * a.H;
  const int q_wave = (lid - bh * n_qt) * QB + wave * 32;
  const int qrow = q_wave + l31;

  // Q fragments (B operand of S^T): lane holds Q[q = l31][dim = 16 ks + 8 hh + j], pre-multiplied by scale * log2(e)
  const float c_log2 = a.scale * LOG2E;
  half8 qf[4];
  {
    const int qc = qrow < a.nq ? qrow : a.nq - 1;
    const half_t* qp = a.Q + (long)b * a.q_bs + (long)qc * a.q_rs + h * 64 + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const half8 raw = *reinterpret_cast<const half8*>(qp + ks * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[ks][j] = (half_t)((float)raw[j] * c_log2);
    }
  }
  const int nkt = (a.nk + KT - 1) / KT;
  // K / V of this (batch, head) through buffer descriptors: rows past nk read as ZERO (range check, no clamping), the
  // per-lane byte offset is fixed for the whole kernel and the tile advances through the SCALAR offset -- no vector
  // instruction is spent on DMA addresses inside the loop
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.K + (long)b * a.k_bs + h * 64), 0,
                                                                      (int)(((long)(a.nk - 1) * a.k_rs + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.V + (long)b * a.v_bs + h * 64), 0,
                                                                      (int)(((long)(a.nk - 1) * a.v_rs + 64) * 2), 0x00020000);
  unsigned dk[8 / NW], dv[8 / NW];
#pragma unroll
  for (int i = 0; i < 8 / NW; ++i) {  // 8 pieces of 8 key rows per tile and operand, spread over the workgroup's waves
    const int r = (wave * (8 / NW) + i) * 8 + (lane >> 3);
    dk[i] = (unsigned)(r * a.k_rs * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4));
    dv[i] = (unsigned)(r * a.v_rs * 2 + (((lane & 7) ^ ((r & 2) << 1)) << 4));
  }
  const unsigned k_tile_bytes = (unsigned)(KT * a.k_rs * 2), v_tile_bytes = (unsigned)(KT * a.v_rs * 2);
  auto stage = [&](int buf, int kt) {
    half_t* Kt = lds + buf * (2 * TILE);
    half_t* Vt = Kt + TILE;
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
      const int rbase = (wave * (8 / NW) + i) * 8;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (WCA_LDS void*)(Kt + rbase * 64), 16, (int)dk[i], (int)(kt * k_tile_bytes), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (WCA_LDS void*)(Vt + rbase * 64), 16, (int)dv[i], (int)(kt * v_tile_bytes), 0, 0);
    }
  };

  // per-lane LDS byte addresses (ring slot 0; the slot and the fragment index go into the instructions' immediate offsets)
  // K fragment (kb, ks): key = 32 kb + l31, chunk (2 ks + hh) ^ ((key >> 1) & 7): bit 0 = hh ^ (kswz & 1), bits 1-2 = ks ^ (kswz >> 1)
  const int kswz = (l31 >> 1) & 7;
  const unsigned lds_base = lds_off(lds);
  unsigned ka[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ka[ks] = lds_base + (unsigned)(l31 * 128 + 16 * ((hh ^ kswz) & 1) + 32 * (ks ^ (kswz >> 1)));
  // V^T fragment (db, s): two transposed reads of 4 keys x 16 dims per 16-lane group:
  //   group g = lane >> 4: lane half hh = g >> 1, dims 32 db + 16 (g & 1) + 4 p .. +3, key 16 s + 4 hh + q (+ 8 for the second read)
  const int vq = (lane & 15) >> 2, vp = lane & 3, vg1 = (lane >> 4) & 1;
  const int vkey0 = 4 * hh + vq;                       // key inside a 16-key step (first read); + 8 for the second
  const int vdim0 = 16 * vg1 + 4 * vp;                 // dim inside a 32-dim block
  // chunk of (db, dim) = 4 db + (vdim0 >> 3); the V swizzle flips chunk bit 2 with key bit 1 (= bit 1 of vkey0)
  const int vsw = (vkey0 & 2) << 1;
  unsigned va[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
    va[db] = lds_base + (unsigned)(2 * TILE + vkey0 * 128 + (((4 * db + (vdim0 >> 3)) ^ vsw) << 4) + 2 * (vdim0 & 7));  // V tile = slot + 8 KiB

  f32x16 ot[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
  half8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.0f;
  float m_run = -INFINITY;  // running row maximum in the scaled log2 domain
  float l_run = 0.f;
  // C operand of the first S^T MFMA of every key block: -m_running in all 16 registers. It lives across tiles and is
  // rewritten only when a row maximum is raised (an accumulator initialised per tile costs 32 v_mov per wave-tile)
  f32x16 cinit;
#pragma unroll
  for (int r = 0; r < 16; ++r) cinit[r] = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // becomes the MFMA's inline 0 operand

  stage(0, 0);
  if (nkt > 1) stage(1, 1);
#define WCA_STAMP(IDX)                                                                      \
  do {                                                                                     \
    if (STAMP) {                                                                           \
      unsigned long long t_;                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      if (lane == 0 && blockIdx.x < 4 && kt < 32 && wave < 4) a.dbg[((blockIdx.x * 4 + wave) * 32 + kt) * 8 + (IDX)] = t_; \
    }                                                                                      \
  } while (0)

  // one 64-key tile in ring slot SLOT (compile time: the LDS offsets of all 24 fragment reads are immediates)
  auto tile = [&](auto slot_c, int kt) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int SB = SLOT * SLOT_BYTES;
    WCA_STAMP(0);
    if (kt + 1 < nkt) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // this wave's four requests of tile kt + 1 stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    WCA_STAMP(1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WCA_STAMP(2);
    if (kt + 2 < nkt) stage((SLOT + 2) % 3, kt + 2);

    // ---- S^T: st[kb][r] = s'(q = l31, key = 64 kt + 32 kb + (r&3) + 8 (r>>2) + 4 hh) - m_running
    // all eight K fragments are requested at once; each key block's MFMAs start when its four have landed
    f32x16 st[2];
    {
      half8 k0[4], k1[4];
      k0[0] = lds_read_b128_asm<SB>(ka[0]);
      k0[1] = lds_read_b128_asm<SB>(ka[1]);
      k0[2] = lds_read_b128_asm<SB>(ka[2]);
      k0[3] = lds_read_b128_asm<SB>(ka[3]);
      k1[0] = lds_read_b128_asm<SB + 32 * 128>(ka[0]);
      k1[1] = lds_read_b128_asm<SB + 32 * 128>(ka[1]);
      k1[2] = lds_read_b128_asm<SB + 32 * 128>(ka[2]);
      k1[3] = lds_read_b128_asm<SB + 32 * 128>(ka[3]);
      WCA_LGKM_WAIT4(4, k0[0], k0[1], k0[2], k0[3]);
      st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0[0], qf[0], cinit, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0[ks], qf[ks], st[0], 0, 0, 0);
      WCA_LGKM_WAIT4(0, k1[0], k1[1], k1[2], k1[3]);
      st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1[0], qf[0], cinit, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1[ks], qf[ks], st[1], 0, 0, 0);
    }
    // V^T fragments of the first 32 output dims: requested now, they land under the softmax
    // (immediate offsets: key step s4 -> + 16 s4 keys x 128 B; the second read of a step + 8 keys x 128 B)
    half4 v0a[4], v0b[4], v1a[4], v1b[4];
    v0a[0] = lds_read_tr_asm<SB + 0 * 2048>(va[0]);
    v0b[0] = lds_read_tr_asm<SB + 0 * 2048 + 1024>(va[0]);
    v0a[1] = lds_read_tr_asm<SB + 1 * 2048>(va[0]);
    v0b[1] = lds_read_tr_asm<SB + 1 * 2048 + 1024>(va[0]);
    v0a[2] = lds_read_tr_asm<SB + 2 * 2048>(va[0]);
    v0b[2] = lds_read_tr_asm<SB + 2 * 2048 + 1024>(va[0]);
    v0a[3] = lds_read_tr_asm<SB + 3 * 2048>(va[0]);
    v0b[3] = lds_read_tr_asm<SB + 3 * 2048 + 1024>(va[0]);
    WCA_STAMP(3);
    // keys past nk (last tile only): wave-uniform branch
    if (kt * KT + KT > a.nk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          st[kb][r] = key >= a.nk ? -INFINITY : st[kb][r];
        }
    }
    // row maximum of (s' - m_eff): 15 v_max3 + 1 v_max over the lane's 32 values, then the partner half (lane ^ 32)
    float mx;
    {
      float m0 = max3f(st[0][0], st[0][1], st[0][2]), m1 = max3f(st[0][3], st[0][4], st[0][5]);
      float m2 = max3f(st[0][6], st[0][7], st[0][8]), m3 = max3f(st[0][9], st[0][10], st[0][11]);
      m0 = max3f(m0, st[0][12], st[0][13]);
      m1 = max3f(m1, st[0][14], st[0][15]);
      m2 = max3f(m2, st[1][0], st[1][1]);
      m3 = max3f(m3, st[1][2], st[1][3]);
      m0 = max3f(m0, st[1][4], st[1][5]);
      m1 = max3f(m1, st[1][6], st[1][7]);
      m2 = max3f(m2, st[1][8], st[1][9]);
      m3 = max3f(m3, st[1][10], st[1][11]);
      m0 = max3f(m0, st[1][12], st[1][13]);
      m1 = max3f(m1, st[1][14], st[1][15]);
      mx = fmaxf(max3f(m0, m1, m2), m3);
      mx = xor32_max(mx);
    }
    // the maximum is raised when the tile maximum of (s' - m_eff) exceeds RESCALE_THR, or anything finite arrives while m_run is still -inf
    if (__any((mx > RESCALE_THR) || (m_run == -INFINITY && mx != -INFINITY))) {
      const float m_eff = (m_run == -INFINITY) ? 0.f : m_run;
      const float m_new = fmaxf(m_run, mx + m_eff);
      const float delta = (m_new == -INFINITY) ? 0.f : m_new - m_eff;  // still to be subtracted from this tile's scores
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
    // p = exp2(s' - m); P^T fragment of k-step s (16 keys): registers 8 (s&1) .. +7 of key block s >> 1
    half8 pf[4];
    {
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = __builtin_amdgcn_exp2f(st[s4 >> 1][8 * (s4 & 1) + j]);
          pf[s4][j] = (half_t)p;
          if (VSUM) ps[j & 3] += p;
        }
      if (VSUM) {
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
      } else {
        // row sums on the matrix pipe: D[i][q] = sum_k P^T[k][q] for every i (the f16-rounded probabilities that enter P.V)
        f32x16 rs = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf[0], zero16, 0, 0, 0);
#pragma unroll
        for (int s4 = 1; s4 < 4; ++s4) rs = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf[s4], rs, 0, 0, 0);
        l_run += rs[0];
      }
      WCA_STAMP(4);
      // ---- O^T += V^T P^T: A operand element j of lane half hh = V[key 16 s + 8 (j>>2) + 4 hh + (j&3)][d = 32 db + l31]
      WCA_LGKM_WAIT8(0, v0a[0], v0b[0], v0a[1], v0b[1], v0a[2], v0b[2], v0a[3], v0b[3]);
      v1a[0] = lds_read_tr_asm<SB + 0 * 2048>(va[1]);
      v1b[0] = lds_read_tr_asm<SB + 0 * 2048 + 1024>(va[1]);
      v1a[1] = lds_read_tr_asm<SB + 1 * 2048>(va[1]);
      v1b[1] = lds_read_tr_asm<SB + 1 * 2048 + 1024>(va[1]);
      v1a[2] = lds_read_tr_asm<SB + 2 * 2048>(va[1]);
      v1b[2] = lds_read_tr_asm<SB + 2 * 2048 + 1024>(va[1]);
      v1a[3] = lds_read_tr_asm<SB + 3 * 2048>(va[1]);
      v1b[3] = lds_read_tr_asm<SB + 3 * 2048 + 1024>(va[1]);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const half8 vf = __builtin_shufflevector(v0a[s4], v0b[s4], 0, 1, 2, 3, 4, 5, 6, 7);
        ot[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s4], ot[0], 0, 0, 0);
      }
      WCA_LGKM_WAIT8(0, v1a[0], v1b[0], v1a[1], v1b[1], v1a[2], v1b[2], v1a[3], v1b[3]);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const half8 vf = __builtin_shufflevector(v1a[s4], v1b[s4], 0, 1, 2, 3, 4, 5, 6, 7);
        ot[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s4], ot[1], 0, 0, 0);
      }
    }
    WCA_STAMP(5);
  };
  // the ring slot of tile kt is kt % 3: three tiles per trip, every slot a compile-time constant
  for (int kt = 0; kt < nkt; kt += 3) {
    tile(IntC<0>{}, kt);
    if (kt + 1 < nkt) tile(IntC<1>{}, kt + 1);
    if (kt + 2 < nkt) tile(IntC<2>{}, kt + 2);
  }
#undef WCA_STAMP

  // ---- epilogue: ot[db][r] = O[q = l31][d = 32 db + (r&3) + 8 (r>>2) + 4 hh]; the two halves of a row are swapped pairwise
  // (v_permlane32_swap) so that each lane stores 16 contiguous bytes
  const float inv = 1.0f / (VSUM ? xor32_sumf(l_run) : l_run);
  half_t* op = a.O + (long)b * a.o_bs + (long)qrow * a.o_rs + h * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      // groups g = 2 g2 and 2 g2 + 1 (4 dims each, at d = 32 db + 8 g + 4 hh): packed f16 x 4 = 2 dwords per group
      unsigned w[2][2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const half2_ pr = half2_{(half_t)(ot[db][4 * (2 * g2 + e) + 2 * k2] * inv), (half_t)(ot[db][4 * (2 * g2 + e) + 2 * k2 + 1] * inv)};
          w[e][k2] = __builtin_bit_cast(unsigned, pr);
        }
      // lane half 0 keeps group 2 g2 (own dims 0-3) and receives the partner's dims 4-7 of the same group; half 1 ends up
      // with group 2 g2 + 1: [partner's dims 0-3 | own dims 4-7]
      unsigned lo0, lo1, hi0, hi1;
      {
        auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
        lo0 = r0[0];
        hi0 = r0[1];
        lo1 = r1[0];
        hi1 = r1[1];
      }
      if (qrow < a.nq) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4*>(op + 32 * db + 16 * g2 + 8 * hh) = u32x4{lo0, lo1, hi0, hi1};
      }
    }
}


// ------------------------------------------------------------------------------------------------
// One-query attention (greedy-decode steps: nq = 1 per utterance and head). The tile kernel above would spend a
// 128-row MFMA tile on one live row; this is a memory-bound stream over the head's K and V rows (1500 x 128 B each for
// the cross-attention: 393 MB per decoder layer at batch 64) with plain VALU math. One workgroup per (utterance,
// head); its four waves take a quarter of the keys each; inside a wave eight lanes share a key (8 dims of 16 bytes
// per lane), so one wave instruction reads 8 key rows, and each 8-lane group keeps its own online-softmax state
// (m, l, o[8 dims]). Groups and waves are merged at the end (log-sum-exp combine through shuffles, then LDS).
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs a) {
  __shared__ float part[4][66];  // per wave: m, l, o[64]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int grp = lane >> 3, c = lane & 7;
  const int bh = blockIdx.x;
  const int b = bh / a.H, h = bh - b * a.H;
  const float c_log2 = a.scale * LOG2E;
  const half_t* qp = a.Q + (long)b * a.q_bs + h * 64 + c * 8;
  const half8 qh = *reinterpret_cast<const half8*>(qp);
  float q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = (float)qh[j] * c_log2;  // scores directly in the log2 domain
  const half_t* Kb = a.K + (long)b * a.k_bs + h * 64 + c * 8;
  const half_t* Vb = a.V + (long)b * a.v_bs + h * 64 + c * 8;
  const int per_wave = ((a.nk + 3) / 4 + 7) & ~7;  // keys per wave, a multiple of the 8 keys of one step
  const int k_lo = wave * per_wave;
  const int k_hi = (k_lo + per_wave < a.nk) ? k_lo + per_wave : a.nk;
  float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.f;
  constexpr int U = 4;  // key steps in flight per wave (32 keys, 8 KiB of K + V)
  for (int k0 = k_lo; k0 < k_hi; k0 += 8 * U) {
    half8 kf[U], vf[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = k0 + u * 8 + grp;
      live[u] = key < k_hi;
      const int kc = live[u] ? key : a.nk - 1;
      kf[u] = *reinterpret_cast<const half8*>(Kb + (long)kc * a.k_rs);
      vf[u] = *reinterpret_cast<const half8*>(Vb + (long)kc * a.v_rs);
    }
    float sc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d = fmaf(q[j], (float)kf[u][j], d);
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      d += __shfl_xor(d, 4);
      sc[u] = live[u] ? d : -INFINITY;
    }
    float mb = sc[0];
#pragma unroll
    for (int u = 1; u < U; ++u) mb = fmaxf(mb, sc[u]);
    const float m_new = fmaxf(m, mb);
    if (m_new != -INFINITY) {
      const float alpha = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - m_new);
      l *= alpha;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] *= alpha;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float p = __builtin_amdgcn_exp2f(sc[u] - m_new);  // 0 for dead keys
        l += p;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(p, (float)vf[u][j], o[j]);
      }
      m = m_new;
    }
  }
  // merge the 8 key groups of the wave (lanes that hold the same dims: xor 8, 16, 32)
#pragma unroll
  for (int off = 8; off <= 32; off <<= 1) {
    const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
    const float mn = fmaxf(m, m2);
    const float a1 = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn);
    const float a2 = (m2 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = o[j] * a1 + __shfl_xor(o[j], off) * a2;
    m = mn;
  }
  if (lane < 8) {
    if (lane == 0) {
      part[wave][0] = m;
      part[wave][1] = l;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[wave][2 + lane * 8 + j] = o[j];
  }
  __syncthreads();
  if (tid < 64) {  // dim tid of the head
    float mn = part[0][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) mn = fmaxf(mn, part[w][0]);
    float lt = 0.f, ov = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float aw = (part[w][0] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(part[w][0] - mn);
      lt += part[w][1] * aw;
      ov += part[w][2 + tid] * aw;
    }
    a.O[(long)b * a.o_bs + h * 64 + tid] = (half_t)(ov / lt);
  }
}

}  // namespace

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
  if (a.split) return launch_attention_split(a, s);  // reference-precision mode: hi/lo operand pairs, three MFMA passes
  if (a.nq <= 0 || a.B <= 0) return hipSuccess;
  if (a.nk <= 0) return hipErrorInvalidValue;
  if ((a.q_rs % 8) || (a.k_rs % 8) || (a.v_rs % 8) || (a.o_rs % 4)) return hipErrorInvalidValue;
  if (a.cap != nullptr && ((a.cap_ld % 4) != 0 || a.cap_ld < ((a.cap_cols + 3) & ~3))) return hipErrorInvalidValue;
  const bool cap = a.cap != nullptr && a.cap_cols > 0;
  if (a.nq == 1 && !cap && !a.dbg && (!a.causal || a.nk == 1)) {  // greedy-decode steps: the KV cache holds exactly the causal prefix
    hipLaunchKernelGGL(attn_decode_kernel, dim3(a.H * a.B), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  dim3 grid(((a.nq + 127) / 128) * a.H * a.B), block(256);
  const size_t shmem = 3 * 2 * TILE * sizeof(half_t);  // 48 KiB
  // encoder self-attention (no mask, no capture, long rows): the 32x32x16 kernel; a.variant 1 forces the 16x16x32 one, 2 the
  // 32x32x16 one (tests)
  const int variant = a.variant ? a.variant : debug_switch(DBG_ATTN_VARIANT);  // debugging aid
  const bool use32 = !a.causal && !cap && (a.o_rs % 8) == 0 && variant != 1 && (a.nq >= 64 || variant >= 2);
  if (use32) {
    dim3 g32(((a.nq + 127) / 128) * a.H * a.B), b32(256);
    if (a.dbg) hipLaunchKernelGGL((attn32_kernel<true>), g32, b32, shmem, s, a);
    else if (variant == 3) hipLaunchKernelGGL((attn32_kernel<false, true>), g32, b32, shmem, s, a);  // row sums on the vector ALU (experiment)
    else hipLaunchKernelGGL((attn32_kernel<false, false>), g32, b32, shmem, s, a);                  // default: row sums on the matrix pipe
    return hipGetLastError();
  }
  if (a.dbg) {
    hipLaunchKernelGGL((attn_kernel<false, false, true>), grid, block, shmem, s, a);
    return hipGetLastError();
  }
  if (a.causal) {
    if (cap) hipLaunchKernelGGL((attn_kernel<true, true>), grid, block, shmem, s, a);
    else hipLaunchKernelGGL((attn_kernel<true, false>), grid, block, shmem, s, a);
  } else {
    if (cap) hipLaunchKernelGGL((attn_kernel<false, true>), grid, block, shmem, s, a);
    else hipLaunchKernelGGL((attn_kernel<false, false>), grid, block, shmem, s, a);
  }
  return hipGetLastError();
}

}  // namespace wca
