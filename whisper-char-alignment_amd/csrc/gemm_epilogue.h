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
  for (int nt = 0; nt < 4; ++nt) col[nt] = (OUT_MODE == 0) ? ((nt >> 1) * 32 + fg * 8 + (nt & 1) * 4) : (nt * 16 + fg * 4);
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
      for (int j = 0; j < 16; ++j) v[j] = gelu_erf(v[j]);
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
    if (OUT_MODE == 0) {
      half_t* cp = reinterpret_cast<half_t*>(a.C) + coff + nbase;
      if (full_n && ((reinterpret_cast<uintptr_t>(cp + fg * 8) & 15) == 0)) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          half8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (half_t)v[hh * 8 + j];
          *reinterpret_cast<half8*>(cp + hh * 32 + fg * 8) = o;
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nbase + col[nt] + r < a.N) cp[col[nt] + r] = (half_t)v[nt * 4 + r];
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

}  // namespace wca
