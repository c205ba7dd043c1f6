// Log-mel front end on gfx950.  Reference call sites: dataset.py:46-48 / dataset.py:107-109 /
// README.md:101-103 -> whisper.pad_or_trim + whisper.log_mel_spectrogram:
//   zero-pad to 480000 samples, STFT(n_fft 400, hop 160, periodic hann, centre/reflect), |.|^2 of the
//   first 3000 frames, mel filterbank, log10(clamp 1e-10), floor at (global max - 8), (x + 4) / 4.
//
// Pass 1 (one workgroup per frame): windowed frame -> LDS, 201-bin real DFT with the n <-> 400-n
//   fold (even part against cos, odd part against sin: 199 steps instead of 399), power spectrum ->
//   LDS, sparse triangular mel filters, log10; per-utterance running max via ordered-int atomicMax.
//   Frames that only see zero padding are skipped: their value is exactly log10(1e-10) = -10.
// Pass 2: floor/scale, writes the API layout [n_mels][3000] f32 and the time-major f16 image
//   [3002][n_mels] (zero rows at both ends) that the conv-stem GEMM reads as overlapping windows.
#include "kernels.h"
#include "wca_common.h"

namespace wca {

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = 201, NFRAMES = 3000, NSAMP = 480000;

__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

__device__ __forceinline__ int active_frames(int n_samples) {
  // frame t touches audio indices [t*160 - 200, t*160 + 199]; non-zero iff t*160 - 200 < n_samples
  int na = (n_samples + 200 + HOP - 1) / HOP;
  return na < NFRAMES ? na : NFRAMES;
}

__global__ void logmel_init_kernel(unsigned* gmax, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) gmax[i] = f2ord(-10.0f);
}

__global__ __launch_bounds__(256) void logmel_power_kernel(LogMelArgs a) {
  __shared__ float s_e[NFFT / 2 + 1];   // even part  e[n] = s[n] + s[400-n], n = 1..199 ; e[0] = s[0], e[200] = s[200]
  __shared__ float s_o[NFFT / 2 + 1];   // odd part   o[n] = s[n] - s[400-n]
  __shared__ float s_raw[NFFT];
  __shared__ f32x2 s_tw[NFFT];
  __shared__ float s_pow[NBIN + 3];
  __shared__ float s_max[4];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int ns = a.n_samples[b];
  if (t >= active_frames(ns)) return;
  const float* pcm = a.pcm + (long)b * a.pcm_stride;
  for (int n = tid; n < NFFT; n += 256) {
    int idx = t * HOP - NFFT / 2 + n;
    if (idx < 0) idx = -idx;                            // reflect (centre=True)
    if (idx >= NSAMP) idx = 2 * (NSAMP - 1) - idx;
    const float x = (idx < ns) ? pcm[idx] : 0.f;
    s_raw[n] = x * a.window[n];
    s_tw[n] = reinterpret_cast<const f32x2*>(a.twiddle)[n];
  }
  __syncthreads();
  for (int n = tid; n <= NFFT / 2; n += 256) {
    if (n == 0 || n == NFFT / 2) {
      s_e[n] = s_raw[n];
      s_o[n] = 0.f;
    } else {
      s_e[n] = s_raw[n] + s_raw[NFFT - n];
      s_o[n] = s_raw[n] - s_raw[NFFT - n];
    }
  }
  __syncthreads();
  if (tid < NBIN) {
    const int k = tid;
    if (a.precise) {
      // reference-precision mode: f64 accumulation of the f32 products (the reference's FFT is accurate to ~1e-7 of the frame's
      // magnitude; a 199-term f32 sum is not). Twiddles and folded samples stay the f32 values above.
      double re = (double)s_e[0] + ((k & 1) ? -(double)s_e[NFFT / 2] : (double)s_e[NFFT / 2]);
      double im = 0.0;
      int idx = 0;
      for (int n = 1; n < NFFT / 2; ++n) {
        idx += k;
        if (idx >= NFFT) idx -= NFFT;
        const f32x2 tw = s_tw[idx];
        re = fma((double)s_e[n], (double)tw[0], re);
        im = fma((double)s_o[n], (double)tw[1], im);
      }
      const float ref = (float)re, imf = (float)im;  // torch.stft returns complex64; abs()**2 is then f32 arithmetic
      s_pow[k] = ref * ref + imf * imf;
    } else {
      float re = s_e[0] + ((k & 1) ? -s_e[NFFT / 2] : s_e[NFFT / 2]);
      float im = 0.f;
      int idx = 0;
      for (int n = 1; n < NFFT / 2; ++n) {
        idx += k;
        if (idx >= NFFT) idx -= NFFT;
        const f32x2 tw = s_tw[idx];
        re = fmaf(s_e[n], tw[0], re);
        im = fmaf(s_o[n], tw[1], im);
      }
      s_pow[k] = re * re + im * im;
    }
  }
  __syncthreads();
  float lv = -INFINITY;
  if (tid < a.n_mels) {
    const int m = tid;
    const int lo = a.filt_lo[m], hi = a.filt_hi[m];
    const float* fr = a.filters + (long)m * NBIN;
    float acc = 0.f;
    if (a.precise) {  // the reference's `filters @ magnitudes` is an f32 matmul in some blocked order: the f64 sum is within half an ulp of any
      double acc64 = 0.0;
      for (int k = lo; k < hi; ++k) acc64 = fma((double)fr[k], (double)s_pow[k], acc64);
      acc = (float)acc64;
    } else {
      for (int k = lo; k < hi; ++k) acc = fmaf(fr[k], s_pow[k], acc);
    }
    lv = log10f(fmaxf(acc, 1e-10f));
    a.scratch[((long)b * a.n_mels + m) * NFRAMES + t] = lv;
  }
  lv = wave_max(lv);
  if ((tid & 63) == 0) s_max[tid >> 6] = lv;
  __syncthreads();
  if (tid == 0) {
    const float mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    atomicMax(a.gmax + b, f2ord(mx));
  }
}

__global__ __launch_bounds__(256) void logmel_finalize_kernel(LogMelArgs a) {
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * 64;
  const int na = active_frames(a.n_samples[b]);
  const float floorv = ord2f(a.gmax[b]) - 8.0f;
  const float* sc = a.scratch + (long)b * a.n_mels * NFRAMES;
  // API layout [m][t]: threads run along t
  if (a.mel_out) {
    float* mo = a.mel_out + (long)b * a.n_mels * NFRAMES;
    for (int e = threadIdx.x; e < a.n_mels * 64; e += 256) {
      const int m = e >> 6, t = t0 + (e & 63);
      if (t < NFRAMES) {
        const float v = (t < na) ? sc[(long)m * NFRAMES + t] : -10.0f;
        mo[(long)m * NFRAMES + t] = (fmaxf(v, floorv) + 4.0f) * 0.25f;
      }
    }
  }
  // time-major f16 image [t + 1][m]: threads run along m
  if (a.mel_tm) {
    half_t* mt = a.mel_tm + (long)b * (NFRAMES + 2) * a.n_mels_pad;
    for (int e = threadIdx.x; e < a.n_mels * 64; e += 256) {
      const int tt = e / a.n_mels, m = e - tt * a.n_mels;
      const int t = t0 + tt;
      if (t < NFRAMES) {
        const float v = (t < na) ? sc[(long)m * NFRAMES + t] : -10.0f;
        const float y = (fmaxf(v, floorv) + 4.0f) * 0.25f;
        if (a.tm_lo) {
          const HalfPair pr = split_pair(y);
          mt[(long)(t + 1) * a.n_mels_pad + m] = pr.hi;
          mt[(long)(t + 1) * a.n_mels_pad + a.tm_lo + m] = pr.lo;
        } else {
          mt[(long)(t + 1) * a.n_mels_pad + m] = (half_t)y;
        }
      }
    }
  }
}

}  // namespace

hipError_t launch_logmel(const LogMelArgs& a, hipStream_t s) {
  if (a.B <= 0) return hipSuccess;
  if (a.n_mels > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(logmel_init_kernel, dim3((a.B + 63) / 64), dim3(64), 0, s, a.gmax, a.B);
  hipLaunchKernelGGL(logmel_power_kernel, dim3(NFRAMES, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(logmel_finalize_kernel, dim3((NFRAMES + 63) / 64, a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace wca
