/*
 * ORACLE (test infrastructure, never shipped, never measured as the product).
 * Plain-C restatement of openai-whisper's whisper/timing.py `dtw_cpu` + `backtrace`, the code that
 * the reference reaches at timing.py:103 (`dtw(-matrix)` on a CPU tensor => dtw_cpu(x.double())).
 * openai-whisper is an un-vendored, unpinned dependency (README.md:8), absent from /root/reference;
 * the algorithm is restated from its published source (>= v20240930), see SURVEY.md Appendix A.3:
 *   cost/trace are float32 tables, x is float64, column-major loop order (j outer, i inner),
 *   diagonal only if strictly smallest, else up only if strictly smallest, else left.
 * PARITY UNPINNED against upstream itself (no upstream build or fixture exists offline); pinned
 * against hand-derived known answers in tests/test_oracle.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* x: [N][M] float64 row-major (already negated by the caller, like dtw(-matrix)).
 * text_idx/time_idx: capacity N+M. Returns the path length, or -1 on allocation failure. */
int wca_oracle_dtw(const double* x, int N, int M, int32_t* text_idx, int32_t* time_idx) {
  const size_t W = (size_t)M + 1;
  float* cost = (float*)malloc(sizeof(float) * (size_t)(N + 1) * W);
  float* trace = (float*)malloc(sizeof(float) * (size_t)(N + 1) * W);
  if (!cost || !trace) {
    free(cost);
    free(trace);
    return -1;
  }
  for (size_t k = 0; k < (size_t)(N + 1) * W; ++k) {
    cost[k] = INFINITY;
    trace[k] = -1.0f;
  }
  cost[0] = 0.0f;
  for (int j = 1; j <= M; ++j) {
    for (int i = 1; i <= N; ++i) {
      const float c0 = cost[(size_t)(i - 1) * W + (j - 1)];
      const float c1 = cost[(size_t)(i - 1) * W + j];
      const float c2 = cost[(size_t)i * W + (j - 1)];
      float c, t;
      if (c0 < c1 && c0 < c2) {
        c = c0;
        t = 0.0f;
      } else if (c1 < c0 && c1 < c2) {
        c = c1;
        t = 1.0f;
      } else {
        c = c2;
        t = 2.0f;
      }
      cost[(size_t)i * W + j] = (float)(x[(size_t)(i - 1) * M + (j - 1)] + (double)c);
      trace[(size_t)i * W + j] = t;
    }
  }
  /* backtrace */
  for (int j = 0; j <= M; ++j) trace[j] = 2.0f;
  for (int i = 0; i <= N; ++i) trace[(size_t)i * W] = 1.0f;
  int i = N, j = M, n = 0;
  const int cap = N + M;
  while ((i > 0 || j > 0) && n < cap) {
    text_idx[n] = i - 1;
    time_idx[n] = j - 1;
    ++n;
    const float t = trace[(size_t)i * W + j];
    if (t == 0.0f) {
      --i;
      --j;
    } else if (t == 1.0f) {
      --i;
    } else if (t == 2.0f) {
      --j;
    } else {
      break; /* upstream raises ValueError("Unexpected trace[i, j]") */
    }
  }
  /* reverse in place (result = path[::-1]) */
  for (int a = 0, b = n - 1; a < b; ++a, --b) {
    int32_t t0 = text_idx[a];
    text_idx[a] = text_idx[b];
    text_idx[b] = t0;
    t0 = time_idx[a];
    time_idx[a] = time_idx[b];
    time_idx[b] = t0;
  }
  free(cost);
  free(trace);
  return n;
}

/* Reflect-padded sliding median along rows (whisper.timing.median_filter, CPU branch:
 * F.pad(reflect) -> unfold -> sort -> [..., w//2]); returns input unchanged if F <= w//2. */
static int cmp_float(const void* a, const void* b) {
  const float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}
void wca_oracle_median_filter(const float* in, float* out, long rows, int F, int w) {
  const int pad = w / 2;
  float* win = (float*)malloc(sizeof(float) * (size_t)(w > 0 ? w : 1));
  for (long r = 0; r < rows; ++r) {
    const float* x = in + r * F;
    float* y = out + r * F;
    if (F <= pad || w <= 1) {
      for (int f = 0; f < F; ++f) y[f] = x[f];
      continue;
    }
    for (int f = 0; f < F; ++f) {
      for (int k = 0; k < w; ++k) {
        int idx = f - pad + k;
        if (idx < 0) idx = -idx;
        if (idx >= F) idx = 2 * (F - 1) - idx;
        win[k] = x[idx];
      }
      qsort(win, (size_t)w, sizeof(float), cmp_float);
      y[f] = win[w / 2];
    }
  }
  free(win);
}
