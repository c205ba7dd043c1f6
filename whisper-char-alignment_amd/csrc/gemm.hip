// f16 x f16 -> f32 MFMA GEMM for gfx950 (MI355X), used for every projection / MLP / conv-stem
// contraction of the Whisper forward (reference call site: timing.py:58 -> whisper.model Linear/Conv1d).
//
//   C[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] )
//
// Four kernels, chosen by launch_gemm:
//   gemm256p_f16_kernel   persistent, software-pipelined 256 x 256 x 64 tile (8 waves): every large GEMM with a flat A
//                         operand -- the encoder's QKV / out / fc1 / fc2, the fused cross-K/V projection, the logits
//   gemm256_f16_kernel    same tile, two barriers per K tile, pointer DMA: the batch-strided conv-stem operands
//   gemm_f16_kernel       128 x 128 x 64 tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave), two workgroups
//                         per CU: GEMMs with fewer than ~192 big tiles (teacher-forced decoder, small batches)
//   gemm_skinny_f16_kernel  M <= 64 rows, weight streaming without LDS staging: the greedy-decode steps
// All use v_mfma_f32_16x16x32_f16 with fp32 accumulation; the tile kernels stage operands HBM -> LDS by LDS-DMA into an
// XOR-swizzled [row][64] f16 image (swizzle applied on the per-lane SOURCE address, read back with the same XOR).
//
// The MFMA is issued with W as the A operand and the activation rows as the B operand, and the
// W fragment rows are permuted, so that each lane ends up holding 16 CONSECUTIVE output columns
// of one output row: the epilogue stores 32 B (f16) / 64 B (f32) per lane per row.
#include <atomic>

#include "gemm_epilogue.h"
#include <cstdlib>
#include "kernels.h"
#include "wca_common.h"

namespace wca {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_ELEMS = 128 * 64;  // one operand tile (f16 elements)

struct RowPtrs {
  const half_t* p[4];
};

__device__ __forceinline__ half8 ldfrag(const half_t* tile, int r, int c) {
  return *reinterpret_cast<const half8*>(tile + r * 64 + ((c ^ swz128(r)) << 3));
}

// Epilogue shared by both tile shapes. Lane (fr, fg) holds, for m-tile mt, the 16 consecutive outputs
// C[m = mbase + mt*16 + fr][n = nb + nt*4 + r]  (nt, r = 0..3).
template <int OUT_MODE, bool GELU, int MT>
__device__ __forceinline__ void epilogue(const GemmArgs& a, f32x4 (&acc)[MT][4], int mbase, int nb, int fr) {
  float bv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bv[j] = (a.bias != nullptr && nb + j < a.N) ? a.bias[nb + j] : 0.f;
  const bool full_n = (nb + 16 <= a.N);

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mbase + mt * 16 + fr;
    if (m >= a.M) continue;
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[mt][nt][r] + bv[nt * 4 + r];
    if (OUT_MODE != 2 && a.addend != nullptr) {   // pre-activation addend (kernels.h)
      const float* ap = a.addend + (long)m * a.ld_addend + nb;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (nb + j < a.N) v[j] += ap[j];
    }
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
      const float* pp = a.pos + (long)(m % a.pos_period) * a.N + nb;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (nb + j < a.N) v[j] += pp[j];
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
      half_t* cp = reinterpret_cast<half_t*>(a.C) + coff + nb;
      if (full_n && ((reinterpret_cast<uintptr_t>(cp) & 15) == 0)) {
        half8 h0, h1, l0, l1;
        if (OUT_MODE == 4) {
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const Half2Pair p0 = split_pair2(v[j], v[j + 1]), p1 = split_pair2(v[8 + j], v[8 + j + 1]);
            h0[j] = p0.hi[0];
            h0[j + 1] = p0.hi[1];
            l0[j] = p0.lo[0];
            l0[j + 1] = p0.lo[1];
            h1[j] = p1.hi[0];
            h1[j + 1] = p1.hi[1];
            l1[j] = p1.lo[0];
            l1[j + 1] = p1.lo[1];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            h0[j] = (half_t)v[j];
            h1[j] = (half_t)v[8 + j];
          }
        }
        reinterpret_cast<half8*>(cp)[0] = h0;
        reinterpret_cast<half8*>(cp)[1] = h1;
        if (OUT_MODE == 4) {  // lo halves (c_lo is a multiple of 8 elements: the same 16-byte alignment)
          reinterpret_cast<half8*>(cp + a.c_lo)[0] = l0;
          reinterpret_cast<half8*>(cp + a.c_lo)[1] = l1;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (nb + j < a.N) {
            if (OUT_MODE == 4) {
              const HalfPair pr = split_pair(v[j]);
              cp[j] = pr.hi;
              cp[a.c_lo + j] = pr.lo;
            } else {
              cp[j] = (half_t)v[j];
            }
          }
      }
    } else {
      float* cp = reinterpret_cast<float*>(a.C) + coff + nb;
      if (full_n && ((reinterpret_cast<uintptr_t>(cp) & 15) == 0)) {
        f32x4* c4 = reinterpret_cast<f32x4*>(cp);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 o = f32x4{v[q * 4 + 0], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]};
          if (OUT_MODE == 2) o += c4[q];
          c4[q] = o;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (nb + j < a.N) cp[j] = (OUT_MODE == 2) ? cp[j] + v[j] : v[j];
      }
    }
  }
}

template <int OUT_MODE, bool GELU, int SITE>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* lds = reinterpret_cast<half_t*>(smem);
  // layout: [buf][A tile | W tile]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int ntn = (a.N + BN - 1) / BN;
  const int ntm = (a.M + BM - 1) / BM;
  const int nwg = ntm * ntn;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tm = id / ntn, tn = id - tm * ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  // Per-lane source rows for the LDS-DMA staging: 4 A rows and 4 W rows, fixed over the K loop.
  // Wave-instruction i of wave w covers tile rows (w*4+i)*8 .. +7; lane L -> row +(L>>3),
  // LDS chunk position L&7, which holds global chunk (L&7) ^ swz(row).
  const half_t* asrc[4];
  const half_t* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz128(r);
    int gm = m0 + r;
    gm = gm < a.M ? gm : a.M - 1;
    long aoff;
    if (a.a_rows_per_batch > 0) {
      const int b = gm / a.a_rows_per_batch;
      const int t = gm - b * a.a_rows_per_batch;
      aoff = (long)b * a.a_batch_stride + (long)t * a.lda;
    } else {
      aoff = (long)gm * a.lda;
    }
    // split-K (gridDim.y > 1): this workgroup multiplies K slice blockIdx.y and stores a raw partial tile (below)
    const int kbeg = (int)blockIdx.y * (a.K / (int)gridDim.y);
    asrc[i] = a.A + aoff + c * 8 + kbeg;
    int gn = n0 + r;
    gn = gn < a.N ? gn : a.N - 1;
    wsrc[i] = a.W + (long)gn * a.ldw + c * 8 + kbeg;
  }

  auto stage = [&](int buf, int k0) {
    half_t* At = lds + buf * (2 * TILE_ELEMS);
    half_t* Wt = At + TILE_ELEMS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rbase = (wave * 4 + i) * 8;
      glds16(asrc[i] + k0, At + rbase * 64);
      glds16(wsrc[i] + k0, Wt + rbase * 64);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;   // fragment row index
  const int fg = lane >> 4;   // k-chunk group
  // activation rows (B operand): natural order; weight rows (A operand): permuted so that
  // MFMA-row i of n-tile nt is output column (i>>2)*16 + nt*4 + (i&3) of the wave's 64.
  int xrow[4], wrow[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    xrow[t] = wm * 64 + t * 16 + fr;
    wrow[t] = wn * 64 + (fr >> 2) * 16 + t * 4 + (fr & 3);
  }

  const int nk = a.K / BK / (int)gridDim.y;
  stage(0, 0);
  wait_vm0();
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
    const half_t* At = lds + cur * (2 * TILE_ELEMS);
    const half_t* Wt = At + TILE_ELEMS;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 xf[4], wf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        xf[t] = ldfrag(At, xrow[t], ks * 4 + fg);
        wf[t] = ldfrag(Wt, wrow[t], ks * 4 + fg);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
    }
    wait_vm0();
    __syncthreads();
    cur ^= 1;
  }

  if (OUT_MODE == 2 && gridDim.y > 1) {
    // split-K: the raw partial tile of this K slice -> workspace [slice][M][N] f32; splitk_reduce_kernel adds the slices in
    // order, the bias and the residual (deterministic; launch_gemm runs it right behind this kernel)
    GemmArgs p = a;
    p.C = a.sk_part + (long)blockIdx.y * a.M * a.N;
    p.ldc = a.N;
    p.c_rows_per_batch = 0;
    p.bias = nullptr;
    p.pos = nullptr;
    epilogue<1, false, 4>(p, acc, m0 + wm * 64, n0 + wn * 64 + fg * 16, fr);
    return;
  }
  epilogue<OUT_MODE, GELU, 4>(a, acc, m0 + wm * 64, n0 + wn * 64 + fg * 16, fr);
}

// x[m][n] += bias[n] + sum_s part[s][m][n] (slices in order): second half of the split-K residual GEMM
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, int M, int N, const float* __restrict__ bias,
                                                            float* __restrict__ x, int ldc) {
  const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of the [M][N] output
  const int n4 = N >> 2;
  if (i4 >= (long)M * n4) return;
  const int m = (int)(i4 / n4), n = (int)(i4 - (long)m * n4) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(part + (long)m * N + n);
  for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4*>(part + ((long)s * M + m) * N + n);
  if (bias != nullptr) v += *reinterpret_cast<const f32x4*>(bias + n);
  f32x4* xp = reinterpret_cast<f32x4*>(x + (long)m * ldc + n);
  *xp = *xp + v;
}


// ------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 512 threads = 8 waves (2 along M x 4 along N, 128 x 64 outputs per wave),
// one workgroup per CU. LDS: ring of two K tiles, each [A 256x64 | W 256x64] f16 = 64 KiB (128 KiB).
// K tile t+2 is requested (LDS-DMA, 8 x 1 KiB per wave) in the MIDDLE of K tile t, right after the last
// ds_read of the ring slot it overwrites, and is only waited for at the top of K tile t+2 with a COUNTED
// vmcnt (the 8 requests of tile t+3 stay in flight) followed by a raw s_barrier -- the loads never drain
// inside the loop. Two barriers per K tile:
//   B1 (top)   : every wave's DMA for this tile has landed        (RAW; reads come after the barrier)
//   B2 (middle): every wave has finished reading this ring slot   (WAR; the refill is issued after it)
template <int OUT_MODE, bool GELU, int SITE>
__global__ __launch_bounds__(512) void gemm256_f16_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* lds = reinterpret_cast<half_t*>(smem);
  constexpr int TILE256 = 256 * 64;  // one operand K tile (f16 elements)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int ntn = (a.N + 255) / 256;
  const int ntm = (a.M + 255) / 256;
  const int nwg = ntm * ntn;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tm = id / ntn, tn = id - tm * ntn;
  const int m0 = tm * 256, n0 = tn * 256;

  const half_t* asrc[4];
  const half_t* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz128(r);
    int gm = m0 + r;
    gm = gm < a.M ? gm : a.M - 1;
    long aoff;
    if (a.a_rows_per_batch > 0) {
      const int b = gm / a.a_rows_per_batch;
      const int t = gm - b * a.a_rows_per_batch;
      aoff = (long)b * a.a_batch_stride + (long)t * a.lda;
    } else {
      aoff = (long)gm * a.lda;
    }
    asrc[i] = a.A + aoff + c * 8;
    int gn = n0 + r;
    gn = gn < a.N ? gn : a.N - 1;
    wsrc[i] = a.W + (long)gn * a.ldw + c * 8;
  }

  auto stage = [&](int buf, int k0) {
    half_t* At = lds + buf * (2 * TILE256);
    half_t* Wt = At + TILE256;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rbase = (wave * 4 + i) * 8;
      glds16(asrc[i] + k0, At + rbase * 64);
      glds16(wsrc[i] + k0, Wt + rbase * 64);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  int xrow[8], wrow[4];
#pragma unroll
  for (int t = 0; t < 8; ++t) xrow[t] = wr * 128 + t * 16 + fr;
#pragma unroll
  for (int t = 0; t < 4; ++t) wrow[t] = wc * 64 + (fr >> 2) * 16 + t * 4 + (fr & 3);

  const int nk = a.K / BK;
  stage(0, 0);
  if (nk > 1) stage(1, BK);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const half_t* At = lds + cur * (2 * TILE256);
    const half_t* Wt = At + TILE256;
    if (kt + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    half8 wf[2][4], xf[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        wf[ks][t] = ldfrag(Wt, wrow[t], ks * 4 + fg);
        xf[ks][t] = ldfrag(At, xrow[t], ks * 4 + fg);
      }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][nt], xf[ks][mt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);

    half8 xg[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) xg[ks][t] = ldfrag(At, xrow[4 + t], ks * 4 + fg);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) stage(cur, (kt + 2) * BK);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[4 + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][nt], xg[ks][mt], acc[4 + mt][nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }

  epilogue<OUT_MODE, GELU, 8>(a, acc, m0 + wr * 128, n0 + wc * 64 + fg * 16, fr);
}

// Epilogue of the pipelined 256 x 256 kernel (column maps documented at `wsrc_row` there): every store
// instruction writes 64 contiguous bytes per output row (4 lanes x 16 B).
// ------------------------------------------------------------------------------------------------
// Software-pipelined, PERSISTENT 256 x 256 x 64 kernel (flat A): ONE barrier per K tile, placed between the two
// K=32 halves. At that point every wave has issued and retired all its ds_reads of the current ring slot
// (fragments of half 1 are fetched while half 0 multiplies), so the barrier is at once
//   * the WAR guard for refilling this slot with K step s+2 (DMA issued right after it), and
//   * the RAW guard for K step s+1 (each wave's vmcnt(0) precedes the barrier), whose half-0 fragments
//     are then fetched under the MFMAs of half 1.
// The K steps of ALL the tiles a workgroup owns (virtual ids blockIdx.x, + gridDim.x, ...; one workgroup per CU)
// form one continuous stream: the last two K tiles of a tile already fetch K tiles 0 and 1 of the NEXT tile, so a
// tile's epilogue runs with the next tile's operands in flight and its stores drain under the next tile's
// MFMAs. (Measured on the one-tile-per-workgroup form with s_memtime stamps: prologue 8 %, epilogue + store
// drain 15 % of a workgroup's life at K = 1024, nothing overlapping them.)
// LDS image: [256 rows][64 f16], 16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7): depends on
// r & 15 only, so every fragment address is one per-lane base + an immediate. The W rows are PERMUTED AT DMA
// TIME (which W row lands in LDS row nt*16 + i of each 64-row block depends on the output type, see `wsrc_row`),
// so that the epilogue's stores are contiguous while both operands are read with natural row order.
// Operands arrive by LDS-DMA through buffer descriptors: rows past M / N read as zero (no clamping), wave w
// request i covers tile rows i*64 + w*8 .. +7, so ONE per-lane byte offset per operand suffices.
// The tile's 256 bias values arrive in LDS by DMA with its first operands: a global load inside the epilogue would
// have to wait (vmcnt is one in-order counter) for the next tile's operand DMA issued just before it.
// SPLITW (reference-precision operands, A rows = [hi(K) | lo(K)], a.a_lo = K, W the PLAIN [N][K] matrix): the K loop walks
// (k tile 0: hi, lo), (k tile 1: hi, lo), ... -- two steps per W K-tile. The W tile is fetched ONCE (on even steps, into its own
// ring of two slots) and its register fragments are re-used by the odd step: per algorithmic K tile 3 tiles of DMA and 40 fragment
// reads per wave instead of the 4 tiles / 48 reads of the K-doubled call [A_hi | A_lo] [W | W]^T, and no second copy of W.
template <int V>
struct GemmIntC {
  static constexpr int value = V;
};

// STAMP: 0 product; 1 s_memtime stamps (+ the tile wrap below); 2 the tile wrap alone (timing of an L2-resident operand footprint)
template <int OUT_MODE, bool GELU, int SITE, int STAMP = 0, int SPLITW_MODE = 0>
__global__ __launch_bounds__(512) void gemm256p_f16_kernel(GemmArgs a) {
  // SPLITW_MODE: 0 single f16 operands; 1 pair operands, two-slot rings for A and W (round 4); 2 pair operands, THREE A slots + ONE W slot (round 5, below)
  constexpr bool SPLITW = SPLITW_MODE != 0;
  constexpr bool RING3 = SPLITW_MODE == 2;
  // (Forms with TWO W slots behind the three A slots -- all 160 KiB of LDS, the tile's bias in ONE register fetched by ds_bpermute_b32 in the epilogue -- measured and
  //  removed: pair operands with every request in half 0 (6 + 6 per two steps): QKV 1.141 vs 1.112 ms, fc1 1.584 vs 1.552, fc2 1.405 vs 1.338 against this form -- six
  //  requests in one half cost more than 4 + 4 + 4; single f16 operands with the A requests in half 0 and the W requests in half 1: QKV -1.2 %, out-projection -5.7 %, fc1
  //  -1.3 %, fc2 -1.5 % against the two-slot rings, bit-identical -- real, but the f16 mode is the secondary operating point and the form doubles the instantiations.)
  // (Form C, measured and removed: ONLY waves 0-3 -- which win every issue arbitration against their SIMD partners 4-7 and then idle ~680 cycles at the step's
  //  barrier -- issue the in-loop requests, two row groups each: QKV 1.069 vs 1.077 ms, out-projection 0.411 vs 0.409, fc1 1.509 vs 1.478, fc2 1.307 vs 1.297 against
  //  form B: the cost of a request is not the issuing wave's own stall.)
  // (Round-5 experiment forms of the RING3 loop, measured on the fc1 shape and removed: no s_setprio toggles around the MFMA groups 1.488 ms, no toggles + a static
  //  priority for waves 4-7 -- the arbitration losers of every SIMD, the critical path of the step -- 1.500, half 0's fragment reads three per group in groups 0-3
  //  1.509, both 1.500, against 1.497 ms for this loop: nothing.)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* lds = reinterpret_cast<half_t*>(smem);
  constexpr int TILE256 = 256 * 64;
  float* bias_lds = reinterpret_cast<float*>(smem + 4 * TILE256 * sizeof(half_t));  // 2 x 256 floats after the ring
  // OUT_MODE 3 (residual + LayerNorm): gamma | beta of the tile's columns (2 x 256 floats each, double buffered like the
  // bias) and the statistics exchange area of epilogue_ln (2560 floats)
  float* gamma_lds = bias_lds + 512;
  float* beta_lds = gamma_lds + 512;
  float* ln_lds = beta_lds + 512;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int ntn = (a.N + 255) / 256;
  const int ntm = (a.M + 255) / 256;
  const int nwg = ntm * ntn;
  const int G = gridDim.x;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.A), 0, (int)a.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.W), 0, (int)a.w_bytes, 0x00020000);
  // bias: 256 floats per tile, double buffered (tile parity), fetched by LDS-DMA with the tile's first operands.
  // A null bias gives a zero-record descriptor: every lane is out of range and the DMA writes zeros.
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, a.bias != nullptr ? a.N * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ln_gamma), 0, OUT_MODE == 3 ? a.N * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rbeta = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ln_beta), 0, OUT_MODE == 3 ? a.N * 4 : 0, 0x00020000);
  const int l8 = lane >> 3;
  const int rho = wave * 8 + l8;                       // LDS row (mod 64) this lane fills
  const int c0 = (lane & 7) ^ ((rho >> 1) & 7);        // global chunk that lands at chunk position lane & 7
  // W row held by LDS row rho = nt*16 + i (i = MFMA row of n-tile nt). Chosen per output type so that in the
  // epilogue the four lanes (fg = 0..3) that share an output row write 64 CONTIGUOUS bytes per store instruction:
  //   f16 out: lane fg owns columns fg*8 .. +7 and 32 + fg*8 .. +7   -> n = (nt>>1)*32 + (i>>2)*8 + (nt&1)*4 + (i&3)
  //   f32 out: lane fg owns columns nt*16 + fg*4 .. +3 (natural)      -> n = nt*16 + i
  const int nt_r = rho >> 4, i_r = rho & 15;
  const int wsrc_row = (OUT_MODE == 0 || OUT_MODE == 3 || OUT_MODE == 4) ? ((nt_r >> 1) * 32 + (i_r >> 2) * 8 + (nt_r & 1) * 4 + (i_r & 3)) : rho;
  // per-lane byte offsets inside a tile (tile base and K offset are wave-uniform and added per request)
  // diagnostic builds, dbg_wrap_kind bit 2 ("packed sources", timing only -- the products are garbage): every DMA piece reads 1 KiB of CONTIGUOUS
  // global memory (as if the operands were stored tile-packed, [panel][K step][256 rows][64]) instead of 8 rows x 128 bytes lda apart; same bytes per
  // panel, same panels. Answers whether the row-strided source pattern costs anything in the L2 -> LDS path.
  const bool packed_src = STAMP != 0 && (a.dbg_wrap_kind & 4) != 0;
  const unsigned va = packed_src ? (unsigned)(rho * 64 + c0 * 8) * 2u : (unsigned)(rho * a.lda + c0 * 8) * 2u;
  const unsigned vw = packed_src ? (unsigned)(wsrc_row * 64 + c0 * 8) * 2u : (unsigned)(wsrc_row * a.ldw + c0 * 8) * 2u;
  const unsigned sa64 = packed_src ? 64u * 64u * 2u : 64u * a.lda * 2u, sw64 = packed_src ? 64u * 64u * 2u : 64u * a.ldw * 2u;

  // request g (0..7) of a K tile: g>>1 = row block (64 rows), g&1 = operand (A / W); 1 KiB per wave each.
  // ta / tw: byte offset of (tile row 0, K offset) in A / W.
  // LDS ring. Two-slot form: slot b = [A K-tile | W K-tile] at b * 64 KiB. RING3 (pair operands): A K-tiles in THREE slots of 32 KiB at 0 / 32 / 64 KiB and
  // ONE W slot at 96 KiB: a W K-tile is read from LDS only in the even step that first uses it (the odd step re-uses the register fragments), so it is dead
  // after that step's mid barrier and the next one can land in its place; the room goes to a third A slot, so that the A tile of step s + 2 can be requested
  // already in HALF 0 of step s (into the slot of step s - 1) and stays in flight across the step's barrier (counted vmcnt(4)): the DMA requests spread over
  // both halves (4 + 4 instead of 0 + 8) and get 1.3 steps of latency cover instead of 0.5-1 (profiles/r05_gemm_stamps.txt: every wave spent ~360 of
  // ~3 450 cycles per step in the vmcnt wait, and half 1 with its 8 requests took 1 600-2 000 cycles against 770-960 for half 0).
  auto a_slot = [&](int buf) -> half_t* { return RING3 ? lds + buf * TILE256 : lds + buf * (2 * TILE256); };
  auto w_slot = [&](int bufw) -> half_t* { return RING3 ? lds + 3 * TILE256 : lds + bufw * (2 * TILE256) + TILE256; };
  auto stage_one = [&](int buf, int bufw, unsigned ta, unsigned tw, int g) {
    half_t* At = a_slot(buf);
    half_t* Wt = w_slot(bufw);
    const int i = g >> 1;
    // the wave-uniform part stays an opaque SGPR value: one v_add per request instead of eight per-lane
    // induction variables (the loop strength reduction otherwise keeps va + i*sa64 + k in 8 VGPRs and spills)
    unsigned so = ((g & 1) == 0) ? ta + i * sa64 : tw + i * sw64;
    asm volatile("" : "+s"(so));
    if ((g & 1) == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (WCA_LDS void*)(At + (i * 64 + wave * 8) * 64), 16, (int)(va + so), 0, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (WCA_LDS void*)(Wt + (i * 64 + wave * 8) * 64), 16, (int)(vw + so), 0, 0, 0);
  };

  auto stage_bias = [&](int par, int ncol0) {
    if (wave < 4) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (WCA_LDS void*)(bias_lds + par * 256 + wave * 64), 4, (ncol0 + wave * 64 + lane) * 4, 0, 0, 0);
      if (OUT_MODE == 3) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (WCA_LDS void*)(gamma_lds + par * 256 + wave * 64), 4, (ncol0 + wave * 64 + lane) * 4, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbeta, (WCA_LDS void*)(beta_lds + par * 256 + wave * 64), 4, (ncol0 + wave * 64 + lane) * 4, 0, 0, 0);
      }
    }
  };

  const int fr = lane & 15, fg = lane >> 4;
  const int pos0 = fg ^ ((fr >> 1) & 7);
  // element offsets of this lane's fragment row 0 for K half 0 / 1 (half 1 = chunk position ^ 4)
  const int xb0 = (wr * 128 + fr) * 64 + pos0 * 8, xb1 = (wr * 128 + fr) * 64 + (pos0 ^ 4) * 8;
  const int wb0 = (wc * 64 + fr) * 64 + pos0 * 8, wb1 = (wc * 64 + fr) * 64 + (pos0 ^ 4) * 8;

#define WCA_LOAD_HALF(AT, WT, XB, WB, WF, XF)                                                      \
  do {                                                                                             \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) WF[t] = *reinterpret_cast<const half8*>((WT) + (WB) + t * 1024); \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) XF[t] = *reinterpret_cast<const half8*>((AT) + (XB) + t * 1024); \
  } while (0)

  const int nk = SPLITW ? 2 * (a.K / BK) : a.K / BK;   // K steps of one tile
  // byte offset of K step s inside an A row / a W row
  auto a_koff = [&](int s_) -> unsigned {
    if (packed_src) return (unsigned)s_ * (256u * 64u * 2u);   // (K step s of a panel: its own 32 KiB)
    return SPLITW ? (unsigned)(((s_ & 1) * (int)a.a_lo + (s_ >> 1) * BK) * 2) : (unsigned)(s_ * (BK * 2));
  };
  auto w_koff = [&](int s_) -> unsigned {
    if (packed_src) return (unsigned)(SPLITW ? (s_ >> 1) : s_) * (256u * 64u * 2u);
    return SPLITW ? (unsigned)((s_ >> 1) * (BK * 2)) : (unsigned)(s_ * (BK * 2));
  };
  // (A/B experiments that did NOT pay on MI355X and were removed: giving the two wave groups different
  //  fetch/MFMA orders or different DMA-issue windows to break SIMD-partner lockstep; a software L2 prefetch
  //  two tiles ahead of the DMA; a 4- and 5-slot ring of K=32 tiles with the DMA 3-4 tiles ahead (-5 %);
  //  padding the leading dimensions off a power of two; de-phasing the workgroups' start by up to 1/8 tile;
  //  the non-temporal hint (aux = 2) on the W stream to keep the activation panels in L2 (-7...-9 %: the workgroups
  //  that share an n-panel then stop sharing it through L2).
  //  Two re-designs around a ring of FOUR K=32 slots (64-byte LDS rows, chunk ^ ((r>>1)&3), DMA four steps ahead,
  //  counted vmcnt, one barrier per K=32 step), both parity-green, neither faster in the pipeline:
  //   * four waves, one per SIMD, 128x128 per wave with the 256 accumulators in AGPRs (inline-asm in-place MFMA):
  //     s_memtime shows 1316-1400 cycles per 64-MFMA step against 1024 for the MFMAs alone -- a wave's own
  //     ds_read / DMA / SALU instructions do not issue underneath its MFMAs, only another wave's do; QKV +8 %,
  //     fc1 -6 %, fc2 -11 % (each wave runs twice the epilogue), bench unchanged;
  //   * eight waves on the same ring: QKV +14 % in isolation (operands L2 resident), cross-KV -5 %, fc2 -4 %
  //     (64-byte DMA segments on operands that miss L2), bench unchanged.)
  int v = blockIdx.x;
  int id = xcd_remap(v, nwg);
  // logical id -> tile: supertiles of SM m-panels swept n-major, m fastest inside. The 32 workgroups that share an
  // XCD's L2 hold 32 consecutive ids = SM m-panels x 32/SM n-panels, and the following rounds walk the same SM
  // activation panels across the remaining n-panels, so those are re-read from L2 instead of the fabric. SM is
  // chosen in launch_gemm (measured, M = 48000: fc1 870 -> 921-960 TFLOP/s at SM 8-16, QKV +2-3 %; K = 4096 and
  // N = 49152 lose 2-10 % with SM > 1 and keep the row-major order).
  const int SM = a.supertile > 0 ? a.supertile : 1;
  auto tile_of = [&](int t, int& tm, int& tn) {
    const int per = SM * ntn;
    const int st = t / per, r = t - st * per;
    const int rows = (ntm - st * SM) < SM ? (ntm - st * SM) : SM;  // the last supertile may be short
    tn = r / rows;
    tm = st * SM + (r - tn * rows);
  };
  // OUT_MODE 3 (LayerNorm in the epilogue): the N/256 workgroups that hold one 256-row panel wait for each other in the
  // epilogue, so a panel's tiles must be worked on AT THE SAME TIME: round r of the 32 workgroups that share an XCD label
  // (blockIdx % 8) covers PR = 32 / ntn whole panels x all ntn column tiles (the same L2 footprint as a supertile of PR), and
  // the panels are dealt to the XCD labels in contiguous blocks. (With the id order above, panels straddle the XCDs' id
  // ranges: a workgroup then waits for tiles that another XCD reaches a whole launch later.) Correctness does not depend on
  // the placement -- a wait only needs same-round tiles of other workgroups, which never wait on this one's later rounds.
  const int spx = G >> 3;                        // workgroups per XCD label (launch_gemm: G % 8 == 0, spx >= ntn)
  const int PR = spx / ntn;                      // panels per round and label
  const int lab = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pq = ntm >> 3, prm = ntm & 7;
  const int lab_panels = pq + (lab < prm ? 1 : 0), lab_base = lab * pq + (lab < prm ? lab : prm);
  auto tile_of_round = [&](int round, int& tm, int& tn) -> bool {
    tn = slot / PR;
    const int pi = round * PR + (slot - tn * PR);
    tm = lab_base + pi;
    return tn < ntn && pi < lab_panels;
  };
  int tm_, tn_;
  int round = 0;
  if (OUT_MODE == 3) {
    if (!tile_of_round(0, tm_, tn_)) return;  // (wave-uniform: the whole workgroup has no tile)
  } else {
    tile_of(id, tm_, tn_);
  }
  // diagnostic builds (STAMP != 0): tile coordinates taken modulo (dbg_wrap_m, dbg_wrap_n) -- for the OPERAND addresses (every workgroup then
  // walks the same few panels: an L2-resident footprint), for the OUTPUT addresses (stores collide in a few tiles), or both (dbg_wrap_kind 1 / 2 / 0;
  // dbg_wrap_m == 15: the diagnostic instantiation without any wrap)
  auto wrap_opnd = [&](int& tm, int& tn) {
    if (STAMP != 0 && a.dbg_wrap_m > 0 && a.dbg_wrap_m < 15 && (a.dbg_wrap_kind & 3) != 2) {
      tm %= a.dbg_wrap_m;
      tn %= a.dbg_wrap_n;
    }
  };
  auto wrap_out = [&](int& tm, int& tn) {
    if (STAMP != 0 && a.dbg_wrap_m > 0 && a.dbg_wrap_m < 15 && (a.dbg_wrap_kind & 3) != 1) {
      tm %= a.dbg_wrap_m;
      tn %= a.dbg_wrap_n;
    }
  };
  int tma_ = tm_, tna_ = tn_;
  wrap_opnd(tma_, tna_);
  wrap_out(tm_, tn_);
  int m0 = tm_ * 256, n0 = tn_ * 256;
  unsigned ta = (unsigned)(tma_ * 256) * a.lda * 2u, tw = (unsigned)(tna_ * 256) * a.ldw * 2u;

  // (Round-5 experiment, removed: starting the 32 workgroups of an XCD label up to 1/8 ... 1 tile period apart, so that the tile epilogues --
  //  256 KiB of stores per workgroup, issued by all 256 CUs at the same moment when they run in phase -- spread over the others' K loops:
  //  pair fc1 1.543 / 1.543 / 1.537 / 1.551 / 1.553 ms at 0 / 1/8 / 1/4 / 1/2 / 1 period, QKV 1.125 ... 1.132, fc2 1.357 ... 1.445: nothing, then worse.)
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8 w0[4], x0[8], w1[4], x1[8];
  int par = 0;  // bias buffer of the current tile
  stage_bias(0, n0);
#pragma unroll
  for (int g = 0; g < 8; ++g) stage_one(0, 0, ta, tw, g);
  if (nk > 1) {
#pragma unroll
    for (int g = 0; g < 8; ++g)
      if (!SPLITW || (g & 1) == 0) stage_one(1, 1, ta + a_koff(1), tw + w_koff(1), g);  // SPLITW: step 1 multiplies the W tile of step 0
    if (SPLITW) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  WCA_LOAD_HALF(a_slot(0), w_slot(0), xb0, wb0, w0, x0);
  int ring_a = 0;          // RING3: A slot of the current K step (runs on across tile boundaries)

#define WCA_STAMP(IDX)                                                                      \
  do {                                                                                     \
    if (STAMP == 1) {                                                                      \
      unsigned long long t_;                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      __builtin_amdgcn_sched_barrier(0);                                                   \
      if (lane == 0 && blockIdx.x < 4 && v == (int)blockIdx.x) a.dbg[((blockIdx.x * 8 + wave) * 64 + kt) * 8 + (IDX)] = t_; \
    }                                                                                      \
  } while (0)
  for (;;) {
    // the tile after this one (persistent launch only: gridDim.x < nwg needs nk even and >= 2, see launch_gemm)
    const int vn = v + G;
    bool has_next;
    int idn = id;
    int tmn_, tnn_;
    if (OUT_MODE == 3) {
      has_next = tile_of_round(round + 1, tmn_, tnn_);
      if (!has_next) {
        tmn_ = m0 >> 8;
        tnn_ = n0 >> 8;
      }
    } else {
      has_next = vn < nwg;
      idn = has_next ? xcd_remap(vn, nwg) : id;
      tile_of(idn, tmn_, tnn_);
    }
    int tman_ = tmn_, tnan_ = tnn_;
    wrap_opnd(tman_, tnan_);
    wrap_out(tmn_, tnn_);
    const int m0n = tmn_ * 256, n0n = tnn_ * 256;
    const unsigned tan = (unsigned)(tman_ * 256) * a.lda * 2u, twn = (unsigned)(tnan_ * 256) * a.ldw * 2u;
    // one K step; PAR >= 0: the step's parity (= its A ring slot) is a compile-time constant (SPLITW: even = first use of a W tile)
    auto kstep = [&](auto par_c, const int kt) {
      constexpr int PAR = decltype(par_c)::value;
      const int cur = RING3 ? ring_a : (PAR >= 0 ? PAR : (kt & 1));
      const int nxt = RING3 ? (ring_a == 2 ? 0 : ring_a + 1) : (cur ^ 1);   // A slot of step kt + 1
      const int wslot = SPLITW ? ((kt >> 1) & 1) : cur;                 // W ring slot of this step (RING3: one slot, w_slot() ignores it)
      const int wslot_n = SPLITW ? (((kt + 1) >> 1) & 1) : (cur ^ 1);   // ... of step kt + 1
      // this step's W fragments are new (SPLITW, odd step: those of the step before); wave-uniform
      const bool W_FRESH = !SPLITW || (PAR >= 0 ? PAR == 0 : (kt & 1) == 0);
      const bool W_NEXT_FRESH = !SPLITW || (PAR >= 0 ? PAR == 1 : (kt & 1) == 1);
      const half_t* At = a_slot(cur);
      const half_t* Wt = w_slot(wslot);
      // K step kt + 2 (of this tile, or of the next one): its A tile and, on even steps, its W tile are requested during this step
      const bool in_tile2 = kt + 2 < nk;
      const bool more1 = kt + 1 < nk, more2 = in_tile2 || has_next;
      const unsigned ka = (in_tile2 ? ta + a_koff(kt + 2) : tan + a_koff(kt + 2 - nk));
      const unsigned kw = (in_tile2 ? tw + w_koff(kt + 2) : twn + w_koff(kt + 2 - nk));
      const int prv = RING3 ? (ring_a == 0 ? 2 : ring_a - 1) : 0;   // RING3: the A slot of step kt - 1 = of step kt + 2 (every wave is past that step's barrier)
      WCA_STAMP(0);
      // ---- K half 0 (fragments w0/x0 were fetched under the previous step's half 1). The 12 fragment reads of
      // half 1 are issued two at a time between groups of 4 MFMAs.
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[nt], x0[g], acc[g][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (g < 2) {
          if (W_FRESH) {
            w1[2 * g] = *reinterpret_cast<const half8*>(Wt + wb1 + (2 * g) * 1024);
            w1[2 * g + 1] = *reinterpret_cast<const half8*>(Wt + wb1 + (2 * g + 1) * 1024);
          }
        } else if (g < 6) {
          x1[2 * (g - 2)] = *reinterpret_cast<const half8*>(At + xb1 + (2 * (g - 2)) * 1024);
          x1[2 * (g - 2) + 1] = *reinterpret_cast<const half8*>(At + xb1 + (2 * (g - 2) + 1) * 1024);
        }
        // RING3: the A tile of step kt + 2 goes out HERE, in half 0 (the two-slot ring cannot: its target slot is still being read), one request in every
        // second group; half 1 keeps only the W requests of the even steps -- at most 4 DMA requests per half instead of 8 in half 1
        if (RING3 && (g & 1) == 0 && more2) stage_one(prv, 0, ka, kw, g);
        __builtin_amdgcn_sched_barrier(0);
      }
      WCA_STAMP(1);
      // all ds_reads of slot `cur` are retired; K step s+1 has landed. Two-slot rings: it is the only DMA in flight. RING3: the 4 youngest requests are
      // the A tile of step s+2 just requested in half 0 (younger than the W requests of the previous step's half 1) and stay in flight across the barrier.
      if (RING3 && more2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      WCA_STAMP(2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      WCA_STAMP(3);
      // ---- K half 1. Slot `cur` is refilled with K step s+2 (of this tile, or of the next one) and step s+1's
      // half-0 fragments are fetched (within a tile), ONE DMA request and up to two ds_reads per group of 4 MFMAs: 64 back-to-back
      // requests per CU right after the barrier serialise in the memory pipe and delay the MFMAs of the waves that
      // issue last (measured with s_memtime stamps: ~1000 cycles of barrier skew per K tile).
      // SPLITW: the W tile of step s+2 is new only when s is even (it goes to the W slot that step s-1 read last: every wave is
      // past that step's barrier), and the next step's W fragments are re-fetched only when s is odd.
      const int wslot_2 = SPLITW ? (wslot ^ 1) : cur;   // W ring slot of step kt + 2
      const half_t* An = a_slot(nxt);
      const half_t* Wn = w_slot(wslot_n);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[nt], x1[g], acc[g][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (RING3) {
          // the W tile of step kt + 2 into the one W slot (dead since this step's barrier), one request in every second group. (Measured alternatives, both
          // slower: the 4 W requests in groups 4-7 -- fc1 -0.4 % instead of -2.9 % against the two-slot rings; two of the odd steps' A requests moved from half 0
          // into their request-free half 1 -- every shape SLOWER than the two-slot rings: a request right behind the barrier, where all eight waves issue at the
          // same instant, costs far more than one in half 0, where the SIMD partners are ~400 cycles apart.)
          if (W_FRESH && (g & 1) == 1 && more2) stage_one(cur, 0, ka, kw, g);
        } else if (more2 && (W_FRESH || (g & 1) == 0)) {
          stage_one(cur, wslot_2, ka, kw, g);
        }
        if (more1) {
          if (g < 2) {
            if (W_NEXT_FRESH) {
              w0[2 * g] = *reinterpret_cast<const half8*>(Wn + wb0 + (2 * g) * 1024);
              w0[2 * g + 1] = *reinterpret_cast<const half8*>(Wn + wb0 + (2 * g + 1) * 1024);
            }
          } else if (g < 6) {
            x0[2 * (g - 2)] = *reinterpret_cast<const half8*>(An + xb0 + (2 * (g - 2)) * 1024);
            x0[2 * (g - 2) + 1] = *reinterpret_cast<const half8*>(An + xb0 + (2 * (g - 2) + 1) * 1024);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      WCA_STAMP(4);
      if (RING3) ring_a = nxt;
    };
    // (SPLITW with the step parity as a compile-time constant -- the loop unrolled by two -- spills 70-90 VGPRs: both steps' LDS
    //  base addresses stay live; the runtime parity costs two scalar branches per step)
    for (int kt = 0; kt < nk; ++kt) kstep(GemmIntC<-1>{}, kt);

    if (STAMP == 1 && lane == 0 && blockIdx.x < 4 && (v / G) < 12)
      a.dbg[((blockIdx.x * 8 + wave) * 64 + 48 + v / G) * 8 + 0] = __builtin_readcyclecounter();
    {
      // opaque copies: keeps the epilogue's per-lane address arithmetic from being hoisted out of the tile loop
      // (live across the K loop it cost 30 VGPRs and spilled)
      int fr_e = fr, fg_e = fg;
      asm volatile("" : "+v"(fr_e), "+v"(fg_e));
      // the epilogue re-reads its arguments from the kernarg segment (scalar loads) instead of keeping ~20 SGPRs
      // of GemmArgs live across the K loop, where they spilled into VGPR lanes
      typedef __attribute__((address_space(4))) GemmArgs KernArgs;
      const KernArgs* ap = (const KernArgs*)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ap));
      if constexpr (OUT_MODE == 3)
        epilogue_ln<KernArgs>(*ap, acc, m0, m0 >> 8, n0 >> 8, ntn, wr, wc, fr_e, fg_e, bias_lds + par * 256 + wc * 64, gamma_lds + par * 256,
                              beta_lds + par * 256, ln_lds);
      else
        epilogue_wide<OUT_MODE, GELU, KernArgs>(*ap, acc, m0 + wr * 128, n0 + wc * 64, fr_e, fg_e, bias_lds + par * 256 + wc * 64);
    }
    if (STAMP == 1 && lane == 0 && blockIdx.x < 4 && (v / G) < 12)
      a.dbg[((blockIdx.x * 8 + wave) * 64 + 48 + v / G) * 8 + 1] = __builtin_readcyclecounter();
    if (!has_next) break;
    // next tile's bias -> the other buffer: its last readers (the epilogue two tiles back) are behind at least one
    // barrier, and the first mid-tile barrier of the new tile (vmcnt(0) on every wave) makes it visible
    par ^= 1;
    stage_bias(par, n0n);
    v = vn;
    id = idn;
    ++round;
    m0 = m0n;
    n0 = n0n;
    ta = tan;
    tw = twn;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // K tile 0 of the new tile landed before the last mid-tile barrier; its fragments are fetched only now so
    // that they are not live across the epilogue (that cost 13-18 spilled VGPRs)
    WCA_LOAD_HALF(a_slot(RING3 ? ring_a : 0), w_slot(0), xb0, wb0, w0, x0);
  }
#undef WCA_STAMP
#undef WCA_LOAD_HALF
}


// ------------------------------------------------------------------------------------------------
// Skinny GEMM for M <= 64 rows (one greedy-decode step: M = batch): weight streaming, latency-bound. A workgroup owns
// 16 output columns for all rows; its four waves split K into quarters, each multiplying its K range for the four
// 16-row m-tiles with operands loaded straight from global memory into MFMA fragments (no LDS staging: every weight
// byte is used once, the <= 64 x K activation block is L2-resident and shared by all workgroups); chunks of 128 K
// values are double-buffered in registers so the next chunk's 20 loads are in flight under the current MFMAs. The
// four partial accumulators are summed through LDS and each wave finishes one m-tile (bias, GELU, f16 / f32 / += f32).
// Grid = N / 16 workgroups (192-256 for the decoder's N = 3072 / 4096, 64 for N = 1024) instead of the 8-32 of the
// 128 x 128 tile kernel.
template <int OUT_MODE, bool GELU>
__global__ __launch_bounds__(256) void gemm_skinny_f16_kernel(GemmArgs a) {
  __shared__ float red[4][4][4][64];  // [wave][m-tile][reg][lane]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int kq = a.K >> 2;            // K range of one wave (a multiple of 128, checked by launch_gemm)
  const int kbeg = wave * kq;
  int nrow = n0 + fr;
  nrow = nrow < a.N ? nrow : a.N - 1;  // columns past N: duplicated weights, results not stored
  const half_t* wp = a.W + (long)nrow * a.ldw + kbeg + fg * 8;
  const half_t* xp[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    int m = mt * 16 + fr;
    m = m < a.M ? m : a.M - 1;
    xp[mt] = a.A + (long)m * a.lda + kbeg + fg * 8;
  }
  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  half8 wf[2][4], xf[2][4][4];
  auto load_chunk = [&](int buf, int k) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      wf[buf][u] = *reinterpret_cast<const half8*>(wp + k + u * 32);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) xf[buf][u][mt] = *reinterpret_cast<const half8*>(xp[mt] + k + u * 32);
    }
  };
  auto mul_chunk = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[buf][u], xf[buf][u][mt], acc[mt], 0, 0, 0);
  };
  load_chunk(0, 0);
  for (int k = 0; k < kq; k += 256) {
    if (k + 128 < kq) load_chunk(1, k + 128);
    mul_chunk(0);
    if (k + 256 < kq) load_chunk(0, k + 256);
    if (k + 128 < kq) mul_chunk(1);
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][mt][r][lane] = acc[mt][r];
  __syncthreads();
  // wave w finishes m-tile w: lane holds C[m = w*16 + fr][n = n0 + 4*fg + r]
  const int m = wave * 16 + fr;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = red[0][wave][r][lane] + red[1][wave][r][lane] + red[2][wave][r][lane] + red[3][wave][r][lane];
  if (m >= a.M) return;
  const int nb = n0 + 4 * fg;
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
}

}  // namespace

bool gemm_ln_supported(int M, int N, int K, int n_cu) {
  if (N % 256 != 0 || N > 2048 || K % 128 != 0 || M < 1) return false;  // whole tiles across the row; an even number of K tiles
  const long tiles = (long)((M + 255) / 256) * (N / 256);
  if (tiles < 192) return false;       // launch_gemm sends fewer tiles to the 128 x 128 kernel
  return (n_cu >> 3) >= N / 256;       // grid = CUs (a multiple of 8): one workgroup per CU, all resident; a round of the n_cu / 8
                                       // workgroups of an XCD label holds at least one whole panel (N / 256 tiles)
}

bool gemm_splitw_supported(int M, int N, int K, int lda, int out_mode) {
  if (M < 1 || N < 1 || K < 128 || (K % 128) != 0) return false;   // whole pairs of K tiles: the W ring's slot parity carries over a tile boundary
  if (out_mode != 0 && out_mode != 1 && out_mode != 2 && out_mode != 4) return false;
  const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  if (tiles256 < 192) return false;                                 // launch_gemm sends fewer tiles to the 128 x 128 kernel
  const size_t a_need = ((size_t)(M - 1) * lda + 2 * (size_t)K) * sizeof(half_t), w_need = ((size_t)(N - 1) * K + K) * sizeof(half_t);
  return a_need < 0x7fffffffull && w_need < 0x7fffffffull;          // buffer-descriptor ranges
}

hipError_t launch_gemm(const GemmArgs& a_in, hipStream_t s) {
  GemmArgs a = a_in;
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (a.K <= 0 || (a.K % BK) != 0) return hipErrorInvalidValue;
  if ((a.lda % 8) != 0 || (a.ldw % 8) != 0) return hipErrorInvalidValue;  // 16-byte LDS-DMA source chunks
  // M <= 64 (greedy-decode steps): weight-streaming skinny kernel; force_tile 64 forces it, 128 etc. bypass it
  if (a.out_mode == 4 && (a.c_lo <= 0 || (a.c_lo & 7))) return hipErrorInvalidValue;
  if (a.addend != nullptr && (a.out_mode == 2 || a.out_mode == 3 || a.a_lo > 0)) return hipErrorInvalidValue;   // (accumulating modes take the extra term as a second
                                                                                                                // accumulating launch; with an addend the pair product is the K-doubled call)
  if ((a.force_tile == 64 || a.force_tile == 0) && a.M <= 64 && (a.K % 512) == 0 && a.a_rows_per_batch == 0 && a.pos == nullptr && a.addend == nullptr && a.out_mode != 4 && a.a_lo <= 0) {
    const dim3 sgrid((unsigned)((a.N + 15) / 16)), sblock(256);
#define WCA_LAUNCH_SK(OM, G) hipLaunchKernelGGL((gemm_skinny_f16_kernel<OM, G>), sgrid, sblock, 0, s, a)
    if (a.out_mode == 0) {
      if (a.gelu) WCA_LAUNCH_SK(0, true); else WCA_LAUNCH_SK(0, false);
    } else if (a.out_mode == 1) {
      if (a.gelu) WCA_LAUNCH_SK(1, true); else WCA_LAUNCH_SK(1, false);
    } else if (a.out_mode == 2 && !a.gelu) {
      WCA_LAUNCH_SK(2, false);
    } else {
      return hipErrorInvalidValue;
    }
#undef WCA_LAUNCH_SK
    return hipGetLastError();
  }
  if (a.force_tile == 64) return hipErrorInvalidValue;
  // tile choice: the 256^2 kernel runs one workgroup per CU, so it needs about a full wave of 256 workgroups
  const long tiles256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  const bool splitw = a.a_lo > 0;
  if (splitw && (a.force_tile == 64 || a.force_tile == 128 || a.force_tile == 256 || a.a_rows_per_batch != 0 ||
                 !gemm_splitw_supported(a.M, a.N, a.K, a.lda, a.out_mode) || (a.a_lo & 7) != 0))
    return hipErrorInvalidValue;
  const size_t a_need = ((size_t)(a.M - 1) * a.lda + (splitw ? (size_t)a.a_lo : 0) + a.K) * sizeof(half_t), w_need = ((size_t)(a.N - 1) * a.ldw + a.K) * sizeof(half_t);
  const bool can_buf = a.a_rows_per_batch == 0 && a_need < 0x7fffffffull && w_need < 0x7fffffffull;
  if (a.a_bytes == 0) a.a_bytes = (unsigned)a_need;
  if (a.w_bytes == 0) a.w_bytes = (unsigned)w_need;
  int dev = 0;
  {
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
  }
  static std::atomic<int> n_cu_cache[32] = {};  // CUs per device ordinal
  int n_cu = n_cu_cache[dev & 31].load(std::memory_order_relaxed);
  if (n_cu == 0) {
    hipError_t e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    n_cu_cache[dev & 31].store(n_cu, std::memory_order_relaxed);
  }
  if (a.cu_limit > 0 && a.cu_limit < n_cu) n_cu = a.cu_limit;   // a CU-masked stream: one persistent workgroup per CU it owns
  // pair operands on the persistent kernel: three A slots + one W slot (round 5); the switch gemm_ring = 1 keeps round 4's two-slot rings (A/B, tests)
  const int ring = splitw ? (debug_switch(DBG_GEMM_RING) == 1 ? 1 : 2) : 0;
  const bool want_big = (a.force_tile == 256 || a.force_tile == 257 || a.force_tile == 258) || (a.force_tile == 0 && tiles256 >= 192);
  const bool pipelined = want_big && can_buf && a.force_tile != 256 && a.addend == nullptr;   // (the pre-activation addend lives in the generic epilogue only: the
                                                                                                 //  persistent kernel's epilogue stays as it is -- an addend launch takes the two-barrier 256 x 256 kernel)
  const bool big = want_big;
  dim3 grid, block;
  size_t shmem;
  int splitk = 1;
  if (big) {
    grid = dim3((unsigned)tiles256);
    block = dim3(512);
    shmem = 2 * 2 * 256 * 64 * sizeof(half_t);  // 128 KiB
    if (pipelined) {
      shmem += 2 * 256 * sizeof(float);  // the tile's bias values, double buffered
      // persistent: one workgroup per CU walks tiles blockIdx.x, + gridDim.x, ... (the ring-slot parity carries
      // over a tile boundary only for an even number of K tiles); force_tile 258 = one tile per workgroup
      const int nk = (splitw ? 2 : 1) * (a.K / BK);
      if (a.force_tile != 258 && nk >= 2 && (nk & 1) == 0 && tiles256 > n_cu) grid = dim3((unsigned)n_cu);
      // (K <= 2048 since round 3: the K-doubled QKV / fc1 of the split mode measure -5 % / -3 % with the supertile order, same-box A/B)
      if (a.supertile <= 0) {
        const int sw = debug_switch(DBG_GEMM_SUPERTILE);   // (tile-order experiments)
        a.supertile = sw > 0 ? sw : (((splitw ? 2 : 1) * a.K <= 2048 && (a.N + 255) / 256 <= 32) ? 8 : 1);
      }
    }
  } else {
    const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
    grid = dim3(ntn * ntm);
    block = dim3(256);
    shmem = 2 * 2 * TILE_ELEMS * sizeof(half_t);  // 64 KiB
    // few tiles and a long K (fc2 of a one- or two-utterance batch: 96 tiles x 64 K tiles): split K over up to 4 workgroups
    // per tile; partial tiles go to the caller's workspace and a second kernel adds them in order (deterministic)
    splitk = 1;
    if (a.out_mode == 2 && !a.gelu && a.sk_part != nullptr && a.pos == nullptr && a.c_rows_per_batch == 0 && (a.N % 4) == 0 && (a.ldc % 4) == 0 &&
        ntn * ntm <= n_cu / 2 && a.K >= 2048) {
      for (int sk = 4; sk >= 2; --sk)
        if (a.K % (sk * BK) == 0 && (size_t)sk * a.M * a.N * sizeof(float) <= a.sk_bytes) {
          splitk = sk;
          break;
        }
    }
    grid.y = (unsigned)splitk;
  }
  // the dynamic-LDS limit is a per-device property of each kernel symbol: remembered per (symbol, device) so that several
  // engines (one per GPU) in one process and concurrent host threads are served correctly
#define WCA_LAUNCH_K(KERN, OM, G, S)                                                              \
  do {                                                                                            \
    static std::atomic<unsigned> attr_mask{0};                                                    \
    if (!(attr_mask.load(std::memory_order_acquire) & (1u << (dev & 31)))) {                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(KERN<OM, G, S>),           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
      if (e != hipSuccess) return e;                                                              \
      attr_mask.fetch_or(1u << (dev & 31), std::memory_order_release);                            \
    }                                                                                             \
    hipLaunchKernelGGL((KERN<OM, G, S>), grid, block, shmem, s, a);                               \
  } while (0)
#define WCA_LAUNCH_K4(KERN, OM, G, S, ST, SW)                                                     \
  do {                                                                                            \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(KERN<OM, G, S, ST, SW>),     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);   \
    if (e != hipSuccess) return e;                                                                \
    hipLaunchKernelGGL((KERN<OM, G, S, ST, SW>), grid, block, shmem, s, a);                       \
  } while (0)
#define WCA_LAUNCH_K5R(KERN, OM, G, S, RING)                                                        \
  do {                                                                                              \
    static std::atomic<unsigned> attr_mask5{0};                                                     \
    if (!(attr_mask5.load(std::memory_order_acquire) & (1u << (dev & 31)))) {                       \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(KERN<OM, G, S, 0, RING>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);   \
      if (e != hipSuccess) return e;                                                                \
      attr_mask5.fetch_or(1u << (dev & 31), std::memory_order_release);                             \
    }                                                                                               \
    hipLaunchKernelGGL((KERN<OM, G, S, 0, RING>), grid, block, shmem, s, a);                        \
  } while (0)
#define WCA_LAUNCH_S(OM, G, S)                            \
  do {                                                    \
    if (pipelined && splitw && ring == 1) { if ((S) == 4) WCA_LAUNCH_K5R(gemm256p_f16_kernel, OM, G, 4, 1); else WCA_LAUNCH_K5R(gemm256p_f16_kernel, OM, G, 1, 1); } \
    else if (pipelined && splitw) { if ((S) == 4) WCA_LAUNCH_K5R(gemm256p_f16_kernel, OM, G, 4, 2); else if ((S) == 3) WCA_LAUNCH_K5R(gemm256p_f16_kernel, OM, G, 3, 2); else if ((S) == 2) WCA_LAUNCH_K5R(gemm256p_f16_kernel, OM, G, 2, 2); else WCA_LAUNCH_K5R(gemm256p_f16_kernel, OM, G, 1, 2); } \
    else if (pipelined) WCA_LAUNCH_K(gemm256p_f16_kernel, OM, G, S); \
    else if (big) WCA_LAUNCH_K(gemm256_f16_kernel, OM, G, S);  \
    else WCA_LAUNCH_K(gemm_f16_kernel, OM, G, S);         \
  } while (0)
#define WCA_LAUNCH(OM, G)                     \
  do {                                        \
    switch (a.site) {                         \
      case 1: WCA_LAUNCH_S(OM, G, 1); break;  \
      case 2: WCA_LAUNCH_S(OM, G, 2); break;  \
      case 3: WCA_LAUNCH_S(OM, G, 3); break;  \
      case 4: WCA_LAUNCH_S(OM, G, 4); break;  \
      default: WCA_LAUNCH_S(OM, G, 0); break; \
    }                                         \
  } while (0)
  // diagnostic launches (tools/gemm_stamps.py; never the product path): s_memtime stamps (a.dbg) and / or the tile wrap, for the encoder's
  // kernel forms only
  if (a.dbg != nullptr || a.dbg_wrap_m > 0) {
    if (!pipelined || a.dbg_wrap_n <= 0 != (a.dbg_wrap_m <= 0)) return hipErrorInvalidValue;
#define WCA_LAUNCH_DIAG(OM, G, SW)                                                      \
  do {                                                                                  \
    if (a.dbg != nullptr) WCA_LAUNCH_K4(gemm256p_f16_kernel, OM, G, 1, 1, SW);          \
    else WCA_LAUNCH_K4(gemm256p_f16_kernel, OM, G, 1, 2, SW);                           \
  } while (0)
    if (splitw && ring == 1) {
      if (a.out_mode == 4 && a.gelu) WCA_LAUNCH_DIAG(4, true, 1);
      else if (a.out_mode == 4) WCA_LAUNCH_DIAG(4, false, 1);
      else if (a.out_mode == 2 && !a.gelu) WCA_LAUNCH_DIAG(2, false, 1);
      else return hipErrorInvalidValue;
    } else if (splitw) {
      if (a.out_mode == 4 && a.gelu) WCA_LAUNCH_DIAG(4, true, 2);
      else if (a.out_mode == 4) WCA_LAUNCH_DIAG(4, false, 2);
      else if (a.out_mode == 2 && !a.gelu) WCA_LAUNCH_DIAG(2, false, 2);
      else return hipErrorInvalidValue;
    } else {
      if (a.out_mode == 0 && a.gelu) WCA_LAUNCH_DIAG(0, true, 0);
      else if (a.out_mode == 0) WCA_LAUNCH_DIAG(0, false, 0);
      else if (a.out_mode == 2 && !a.gelu) WCA_LAUNCH_DIAG(2, false, 0);
      else return hipErrorInvalidValue;
    }
#undef WCA_LAUNCH_DIAG
    return hipGetLastError();
  }
  if (a.out_mode == 3) {
    // residual + LayerNorm epilogue: persistent 256 x 256 kernel only (every workgroup of a 256-row panel must be resident:
    // one workgroup per CU, grid <= CUs), N a multiple of 256; the caller falls back to out_mode 2 + launch_layernorm_f16
    // where gemm_ln_supported() says no
    if (!gemm_ln_supported(a.M, a.N, a.K, n_cu) || a.gelu || !pipelined || a.force_tile == 258 || a.pos != nullptr || a.c_rows_per_batch != 0 || !a.ln_gamma ||
        !a.ln_beta || !a.ln_out || !a.ln_stats || !a.ln_cnt || (a.ldc & 3) || (a.ln_ld & 7))
      return hipErrorInvalidValue;
    const size_t cnt_bytes = (((size_t)((a.M + 255) / 256) * sizeof(unsigned)) + 15) / 16 * 16;
    hipError_t e = hipMemsetAsync(a.ln_cnt, 0, cnt_bytes, s);
    if (e != hipSuccess) return e;
    shmem += (2 * 512 + 2560) * sizeof(float);  // gamma | beta (double buffered) + the statistics exchange area
    grid = dim3((unsigned)(n_cu & ~7));          // round-based panel walk: 8 XCD labels x n_cu / 8 workgroups (idle ones exit)
    switch (a.site) {
      case 4: WCA_LAUNCH_K(gemm256p_f16_kernel, 3, false, 4); break;
      default: WCA_LAUNCH_K(gemm256p_f16_kernel, 3, false, 1); break;
    }
  } else if (a.out_mode == 0) {
    if (a.gelu) WCA_LAUNCH(0, true); else WCA_LAUNCH(0, false);
  } else if (a.out_mode == 1) {
    if (a.gelu) WCA_LAUNCH(1, true); else WCA_LAUNCH(1, false);
  } else if (a.out_mode == 2) {
    if (a.gelu) return hipErrorInvalidValue;
    WCA_LAUNCH(2, false);
  } else if (a.out_mode == 4) {
    if (a.gelu) WCA_LAUNCH(4, true); else WCA_LAUNCH(4, false);
  } else {
    return hipErrorInvalidValue;
  }
#undef WCA_LAUNCH
#undef WCA_LAUNCH_S
#undef WCA_LAUNCH_K
#undef WCA_LAUNCH_K4
#undef WCA_LAUNCH_K5R
  {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (splitk > 1) {
    const long n4 = (long)a.M * (a.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a.sk_part, splitk, a.M, a.N, a.bias,
                       reinterpret_cast<float*>(a.C), a.ldc);
  }
  return hipGetLastError();
}

}  // namespace wca
