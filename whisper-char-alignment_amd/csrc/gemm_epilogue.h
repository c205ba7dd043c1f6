// Epilogue of the 256 x 256 pipelined GEMM kernel (gemm.hip): bias / GELU / positional add /
// f16, f32 or read-modify-write f32 stores of one wave's 128 x 64 accumulator block.
// Lane (fr, fg) holds acc[mt][nt][r] = C[m = mbase + mt*16 + fr][n = nbase + col(nt) + r] where col(nt) is
//   f16 output:  (nt>>1)*32 + fg*8 + (nt&1)*4     (the W rows were permuted at staging time so that the four lanes
//   f32 output:  nt*16 + fg*4                       sharing an output row write 64 contiguous bytes per store)
#pragma once
#include "kernels.h"
#include "wca_common.h"

namespace wca {

// ArgsT: GemmArgs, or GemmArgs in the constant address space (the kernarg segment: fields are then re-read with
// scalar loads where they are used instead of staying live in SGPRs across the caller's K loop).
template <int OUT_MODE, bool GELU, typename ArgsT>
__device__ __forceinline__ void epilogue_wide(const ArgsT& a, f32x4 (&acc)[8][4], int mbase, int nbase, int fr, int fg, const float* bias_l) {
  // column of value (nt, r = 0) relative to nbase
  int col[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) col[nt] = (OUT_MODE == 0 || OUT_MODE == 4) ? ((nt >> 1) * 32 + fg * 8 + (nt & 1) * 4) : (nt * 16 + fg * 4);
  // bias_l: this wave's 64 bias values in LDS (zero where there is no bias / past N)
  float bv[16];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_l + col[nt]);
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[nt * 4 + r] = b4[r];
  }
  const bool full_n = (nbase + 64 <= a.N);
  if (OUT_MODE == 2) {
    // read-modify-write of C: the loads of THREE row groups are in flight together (48 VGPRs, free once the K loop
    // is over); load -> add -> store one row group at a time exposed the full memory latency 8 times per tile
    // (s_memtime: 31-46 k cycles per tile, 38 % of the attention out-projection's life)
    float* cbase = reinterpret_cast<float*>(a.C) + nbase;
    if (full_n && a.pos == nullptr && (a.ldc & 3) == 0 && (a.c_batch_stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(cbase + fg * 4) & 15) == 0)) {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][nt][r] += bv[nt * 4 + r];
      constexpr int NB = 3;  // row groups in flight (4 and 5 fit the register file too and measure the same: with three the
                             // epilogue is already bound by the HBM traffic of the read-modify-write, not by its latency)
#pragma unroll
      for (int base = 0; base < 8; base += NB) {
        f32x4 cv[NB][4];
        long coff[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if (base + j >= 8) continue;
          const int m = mbase + (base + j) * 16 + fr;
          if (a.c_rows_per_batch > 0) {
            const int b = m / a.c_rows_per_batch;
            coff[j] = (long)b * a.c_batch_stride + (long)(m - b * a.c_rows_per_batch) * a.ldc;
          } else {
            coff[j] = (long)m * a.ldc;
          }
          if (m < a.M) {
            const f32x4* cp = reinterpret_cast<const f32x4*>(cbase + coff[j] + fg * 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cv[j][nt] = cp[nt * 4];
          }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if (base + j >= 8) continue;
          const int m = mbase + (base + j) * 16 + fr;
          if (m < a.M) {
            f32x4* cp = reinterpret_cast<f32x4*>(cbase + coff[j] + fg * 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cp[nt * 4] = acc[base + j][nt] + cv[j][nt];
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = mbase + mt * 16 + fr;
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[mt][nt][r] + bv[nt * 4 + r];
    if (m >= a.M) continue;
    if (GELU) {
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        if (OUT_MODE == 4) {   // pair output: the erf GELU to fp32 accuracy, two elements per packed instruction
          const f32x2 g = gelu_erfc2(f32x2{v[j], v[j + 1]});
          v[j] = g[0];
          v[j + 1] = g[1];
        } else {
          v[j] = gelu_erf(v[j]);
          v[j + 1] = gelu_erf(v[j + 1]);
        }
      }
    }
    if (a.pos != nullptr) {
      const float* pp = a.pos + (long)(m % a.pos_period) * a.N + nbase;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nbase + col[nt] + r < a.N) v[nt * 4 + r] += pp[col[nt] + r];
    }
    long coff;
    if (a.c_rows_per_batch > 0) {
      const int b = m / a.c_rows_per_batch;
      const int t = m - b * a.c_rows_per_batch;
      coff = (long)b * a.c_batch_stride + (long)t * a.ldc;
    } else {
      coff = (long)m * a.ldc;
    }
    if (OUT_MODE == 0 || OUT_MODE == 4) {
      half_t* cp = reinterpret_cast<half_t*>(a.C) + coff + nbase;
      if (full_n && ((reinterpret_cast<uintptr_t>(cp + fg * 8) & 15) == 0)) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          half8 o, ol;
          if (OUT_MODE == 4) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              const Half2Pair pr = split_pair2(v[hh * 8 + j], v[hh * 8 + j + 1]);
              o[j] = pr.hi[0];
              o[j + 1] = pr.hi[1];
              ol[j] = pr.lo[0];
              ol[j + 1] = pr.lo[1];
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)v[hh * 8 + j];
          }
          *reinterpret_cast<half8*>(cp + hh * 32 + fg * 8) = o;
          if (OUT_MODE == 4) *reinterpret_cast<half8*>(cp + a.c_lo + hh * 32 + fg * 8) = ol;  // lo halves (c_lo % 8 == 0: same alignment)
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nbase + col[nt] + r < a.N) {
              if (OUT_MODE == 4) {
                const HalfPair pr = split_pair(v[nt * 4 + r]);
                cp[col[nt] + r] = pr.hi;
                cp[a.c_lo + col[nt] + r] = pr.lo;
              } else {
                cp[col[nt] + r] = (half_t)v[nt * 4 + r];
              }
            }
      }
    } else {
      float* cp = reinterpret_cast<float*>(a.C) + coff + nbase;
      if (full_n && ((reinterpret_cast<uintptr_t>(cp + fg * 4) & 15) == 0)) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          f32x4* c4 = reinterpret_cast<f32x4*>(cp + col[nt]);
          f32x4 o = f32x4{v[nt * 4 + 0], v[nt * 4 + 1], v[nt * 4 + 2], v[nt * 4 + 3]};
          if (OUT_MODE == 2) o += *c4;
          *c4 = o;
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nbase + col[nt] + r < a.N) cp[col[nt] + r] = (OUT_MODE == 2) ? cp[col[nt] + r] + v[nt * 4 + r] : v[nt * 4 + r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// OUT_MODE 3: residual add + LayerNorm in the epilogue of the persistent 256 x 256 kernel.
//   x[m][n] += acc + bias        (f32 read-modify-write, as OUT_MODE 2)
//   ln_out[m][n] = (x[m][n] - mean_m) * rstd_m * gamma[n] + beta[n]     (f16; mean / variance over the whole row of N)
// A row spans N / 256 tiles, i.e. N / 256 workgroups (all resident: one per CU, persistent grid). Each computes the
// statistics of its 256 columns from the values it holds in registers -- per wave (64 columns) a two-pass mean / M2,
// merged over the four waves and then over the tiles with Chan's update, so no E[x^2] - mean^2 cancellation -- publishes
// them, and waits for its panel's other tiles; the normalised values are then written straight from the accumulator
// registers. Compared with a separate LayerNorm launch this removes one full read of the f32 residual stream per call.
// Hand-off (cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "Valid forms", first table row): the {mean, M2} pairs
// are 8-byte agent-scope relaxed atomic stores (write-through, sc1), every storing wave drains (vmcnt(0)), the workgroup
// barrier, ONE lane adds to the panel's arrival counter (agent scope); the consumer's ONE wave polls that counter with
// relaxed agent loads, then the workgroup barrier, then every load of the pairs is again an agent-scope (sc1) load.
// Placement-independent; the spin is bounded and raises *ln_err instead of hanging.
__device__ __forceinline__ float xor16_sum(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// merge (mean_a, M2_a) over na values with (mean_b, M2_b) over nb values
__device__ __forceinline__ void chan_merge(float& mean_a, float& m2_a, float na, float mean_b, float m2_b, float nb) {
  const float dlt = mean_b - mean_a, nt = na + nb;
  mean_a += dlt * (nb / nt);
  m2_a += m2_b + dlt * dlt * (na * nb / nt);
}

typedef __attribute__((address_space(1))) unsigned long long gu64_t;
typedef __attribute__((address_space(1))) unsigned gu32_t;

// ln_lds: [256 rows][4 waves] float2 wave partials (8 KiB), then [256] float2 row {mean, rstd} (2 KiB); gb_l: this tile's
// gamma | beta (256 floats each) in LDS. tm / tn: tile coordinates, ntn = N / 256.
template <typename ArgsT>
__device__ __forceinline__ void epilogue_ln(const ArgsT& a, f32x4 (&acc)[8][4], int m0, int tm, int tn, int ntn, int wr, int wc, int fr, int fg,
                                            const float* bias_l, const float* gamma_l, const float* beta_l, float* ln_lds) {
  const int tid = threadIdx.x;
  const int mbase = m0 + wr * 128, nbase = tn * 256 + wc * 64;
  // column of value (nt, r = 0) relative to nbase: the f16 map (W rows were permuted at DMA time): lane fg owns columns
  // fg*8 .. +7 and 32 + fg*8 .. +7, so a row's four lanes write 128 B (f32) / 64 B (f16) contiguous per store group
  int col[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) col[nt] = (nt >> 1) * 32 + fg * 8 + (nt & 1) * 4;
  {
    float bv[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_l + col[nt]);
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[nt * 4 + r] = b4[r];
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][nt][r] += bv[nt * 4 + r];
  }
  // ---- x += acc (three row groups of loads in flight, as in the OUT_MODE 2 epilogue); acc then holds the NEW x
  float* cbase = reinterpret_cast<float*>(a.C) + nbase;
  constexpr int NB = 3;
#pragma unroll
  for (int base = 0; base < 8; base += NB) {
    f32x4 cv[NB][4];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (base + j >= 8) continue;
      const int m = mbase + (base + j) * 16 + fr;
      if (m < a.M) {
        const float* cp = cbase + (long)m * a.ldc;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) cv[j][nt] = *reinterpret_cast<const f32x4*>(cp + col[nt]);
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) cv[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (base + j >= 8) continue;
      const int m = mbase + (base + j) * 16 + fr;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[base + j][nt] += cv[j][nt];
      if (m < a.M) {
        float* cp = cbase + (long)m * a.ldc;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(cp + col[nt]) = acc[base + j][nt];
      }
    }
  }
  // ---- statistics of this wave's 64 columns, per row: two passes over the 16 values a lane holds, summed over the row's 4 lanes
  f32x2* wpart = reinterpret_cast<f32x2*>(ln_lds);          // [256][4]
  f32x2* rowst = reinterpret_cast<f32x2*>(ln_lds + 2048);   // [256]
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[mt][nt][r];
    s = xor32_sum(xor16_sum(s));
    const float mean_w = s * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dv = acc[mt][nt][r] - mean_w;
        q = fmaf(dv, dv, q);
      }
    q = xor32_sum(xor16_sum(q));
    if (fg == 0) wpart[(wr * 128 + mt * 16 + fr) * 4 + wc] = f32x2{mean_w, q};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // ---- tile statistics (256 columns) per row, published for the panel's other tiles
  const int Mpad = ((a.M + 255) / 256) * 256;
  if (tid < 256) {
    f32x2 p0 = wpart[tid * 4 + 0], p1 = wpart[tid * 4 + 1], p2 = wpart[tid * 4 + 2], p3 = wpart[tid * 4 + 3];
    float mean = p0[0], m2 = p0[1];
    chan_merge(mean, m2, 64.f, p1[0], p1[1], 64.f);
    float mean_b = p2[0], m2_b = p2[1];
    chan_merge(mean_b, m2_b, 64.f, p3[0], p3[1], 64.f);
    chan_merge(mean, m2, 128.f, mean_b, m2_b, 128.f);
    const unsigned long long pk = ((unsigned long long)__float_as_uint(m2) << 32) | __float_as_uint(mean);
    __hip_atomic_store((gu64_t*)(a.ln_stats + (long)tn * Mpad + m0 + tid), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores before the signal
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (tid == 0) {
    __hip_atomic_fetch_add((gu32_t*)(a.ln_cnt + tm), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ONE lane polls the panel's counter (relaxed agent loads, bounded); the other waves wait at the barrier below
    unsigned spins = 0;
    while (__hip_atomic_load((gu32_t*)(a.ln_cnt + tm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ntn) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 24)) {  // ~seconds: never reached unless a workgroup of the panel is not running
        if (a.ln_err) atomicOr(a.ln_err, 2);
        break;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the loads below the poll (all of them are sc1)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (tid < 256) {
    float mean = 0.f, m2 = 0.f, nn = 0.f;
    for (int t = 0; t < ntn; ++t) {
      const unsigned long long pk = __hip_atomic_load((gu64_t*)(a.ln_stats + (long)t * Mpad + m0 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float mean_t = __uint_as_float((unsigned)pk), m2_t = __uint_as_float((unsigned)(pk >> 32));
      if (t == 0) {
        mean = mean_t;
        m2 = m2_t;
      } else {
        chan_merge(mean, m2, nn, mean_t, m2_t, 256.f);
      }
      nn += 256.f;
    }
    rowst[tid] = f32x2{mean, __builtin_amdgcn_rsqf(m2 / nn + a.ln_eps)};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // ---- normalise the values still in registers and store them as f16
  float gv[16], bt[16];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma_l + wc * 64 + col[nt]);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta_l + wc * 64 + col[nt]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gv[nt * 4 + r] = g4[r];
      bt[nt * 4 + r] = b4[r];
    }
  }
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = mbase + mt * 16 + fr;
    const f32x2 st = rowst[wr * 128 + mt * 16 + fr];
    if (m >= a.M) continue;
    half_t* op = a.ln_out + (long)m * a.ln_ld + nbase;
#pragma unroll
    for (int hh2 = 0; hh2 < 2; ++hh2) {
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int nt = hh2 * 2 + (j >> 2), r = j & 3;
        o[j] = (half_t)((acc[mt][nt][r] - st[0]) * st[1] * gv[nt * 4 + r] + bt[nt * 4 + r]);
      }
      *reinterpret_cast<half8*>(op + hh2 * 32 + fg * 8) = o;
    }
  }
}

}  // namespace wca
