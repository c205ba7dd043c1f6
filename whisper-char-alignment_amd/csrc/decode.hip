// Greedy ASR pre-pass (reference call sites infer_ali.py:40,60-61 / probe_oracle.py:59-60: `whisper.decode(model,
// mels, DecodingOptions(language="en"))`, upstream openai-whisper decoding.py -- absent third-party dependency,
// restated from its published algorithm): the per-step device kernels of the autoregressive loop.
//   * embed_step:     x[b] = token_embedding[tokens[b][t]] + positional_embedding[t]
//   * kv_append:      self-attention K/V of the new position into the per-layer cache [B][T_max][d]
//   * decode_select:  the logit filters (SuppressBlank, SuppressTokens, ApplyTimestampRules), GreedyDecoder.update
//                     (argmax at temperature 0, log-softmax of the FILTERED logits for sum_logprobs, EOT latching)
//                     for one batch row per workgroup; integer / index work, bit-exact against the oracle on the same
//                     fp32 logits (tests/test_decode_gpu.py).
#include "kernels.h"
#include "wca_common.h"

namespace wca {

namespace {

__global__ __launch_bounds__(256) void embed_step_kernel(const int* __restrict__ tokens, int T_max, int t, const half_t* __restrict__ tok_emb,
                                                         const float* __restrict__ pos_emb, float* __restrict__ x, int d, int n_vocab) {
  const int b = blockIdx.x;
  long tok = tokens[(long)b * T_max + t];
  tok = (tok < 0 || tok >= n_vocab) ? 0 : tok;  // ids come from decode_select / the validated prompt; never read out of bounds
  const half_t* e = tok_emb + tok * d;
  const float* p = pos_emb + (long)t * d;
  float* o = x + (long)b * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) o[c] = (float)e[c] + p[c];
}

// qkv [B][3d] (q | k | v) -> kc / vc [B][T_max][d] at position t (8 halfs per thread)
__global__ __launch_bounds__(256) void kv_append_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ kc, half_t* __restrict__ vc,
                                                        int T_max, int t, int d) {
  const int b = blockIdx.x;
  const half8* k = reinterpret_cast<const half8*>(qkv + (long)b * 3 * d + d);
  const half8* v = reinterpret_cast<const half8*>(qkv + (long)b * 3 * d + 2 * d);
  half8* ko = reinterpret_cast<half8*>(kc + ((long)b * T_max + t) * d);
  half8* vo = reinterpret_cast<half8*>(vc + ((long)b * T_max + t) * d);
  for (int c = threadIdx.x; c < d / 8; c += blockDim.x) {
    ko[c] = k[c];
    vo[c] = v[c];
  }
}

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r += red[i];
  return r;
}

// One workgroup per batch row. cur_len = number of tokens the row holds (the new token is written at index cur_len).
__global__ __launch_bounds__(1024) void decode_select_kernel(DecodeSelectArgs a) {
  __shared__ float red[16];
  __shared__ int red_i[16];
  __shared__ int hist[4];  // last_was_ts, penult_was_ts, have_ts, ts_bound
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const float* lg = a.logits + (long)b * a.ld;
  int* tok = a.tokens + (long)b * a.T_max;
  const int V = a.n_vocab;
  const bool first = (a.cur_len == a.n_initial);

  if (tid == 0) {
    // ApplyTimestampRules bookkeeping over the sampled tokens tokens[n_initial : cur_len] (decoding.py, restated)
    const int ns = a.cur_len - a.n_initial;
    const int* seq = tok + a.n_initial;
    const bool last = ns >= 1 && seq[ns - 1] >= a.timestamp_begin;
    const bool pen = ns < 2 || seq[ns - 2] >= a.timestamp_begin;
    int last_ts = -1;
    for (int i = 0; i < ns; ++i)
      if (seq[i] >= a.timestamp_begin) last_ts = seq[i];
    hist[0] = last;
    hist[1] = pen;
    hist[2] = last_ts >= 0;
    // timestamps must not decrease; they may repeat only when closing a segment
    hist[3] = (last_ts >= 0) ? ((last && !pen) ? last_ts : last_ts + 1) : 0;
  }
  __syncthreads();
  const bool last_ts = hist[0], pen_ts = hist[1], have_ts = hist[2];
  const int ts_bound = hist[3];

  auto dead = [&](int v) -> bool {
    if (a.suppress_mask[v]) return true;  // SuppressTokens (+ <|notimestamps|> under the timestamp rules)
    if (first && a.blank_mask != nullptr && a.blank_mask[v]) return true;  // SuppressBlank
    if (a.apply_timestamp_rules) {
      if (last_ts) {
        if (pen_ts) {
          if (v >= a.timestamp_begin) return true;  // has to be non-timestamp
        } else {
          if (v < a.eot) return true;  // cannot be normal text tokens
        }
      }
      if (have_ts && v >= a.timestamp_begin && v < ts_bound) return true;
      if (first) {
        if (v < a.timestamp_begin) return true;  // must start with a timestamp
        if (a.max_initial_timestamp_index >= 0 && v > a.timestamp_begin + a.max_initial_timestamp_index) return true;
      }
    }
    return false;
  };

  // pass 1: maxima over the filtered logits (all / text part)
  float m_all = -INFINITY, m_text = -INFINITY;
  for (int v = tid; v < V; v += blockDim.x) {
    if (dead(v)) continue;
    const float x = lg[v];
    m_all = fmaxf(m_all, x);
    if (v < a.timestamp_begin) m_text = fmaxf(m_text, x);
  }
  m_all = block_max(m_all, red);
  m_text = block_max(m_text, red);
  // pass 2: partition sums relative to m_all
  float s_text = 0.f, s_ts = 0.f;
  for (int v = tid; v < V; v += blockDim.x) {
    if (dead(v)) continue;
    const float e = __expf(lg[v] - m_all);
    if (v < a.timestamp_begin) s_text += e; else s_ts += e;
  }
  s_text = block_sum(s_text, red);
  s_ts = block_sum(s_ts, red);
  // "if sum of probability over timestamps is above any other token, sample timestamp":
  // logsumexp(logprobs[ts:]) > max(logprobs[:ts])  <=>  log(s_ts) > m_text - m_all
  bool text_dead = false;
  if (a.apply_timestamp_rules && s_ts > 0.f && logf(s_ts) > m_text - m_all) text_dead = true;
  // pass 3: argmax (lowest index among equals) of the final filtered logits
  float best = -INFINITY;
  int best_i = 0x7fffffff;
  for (int v = tid; v < V; v += blockDim.x) {
    if (dead(v) || (text_dead && v < a.timestamp_begin)) continue;
    const float x = lg[v];
    if (x > best) {
      best = x;
      best_i = v;
    }
  }
  {
    // wave then block arg-reduction
    for (int off = 32; off >= 1; off >>= 1) {
      const float ob = __shfl_xor(best, off);
      const int oi = __shfl_xor(best_i, off);
      if (ob > best || (ob == best && oi < best_i)) {
        best = ob;
        best_i = oi;
      }
    }
    const int w = tid >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((tid & 63) == 0) {
      red[w] = best;
      red_i[w] = best_i;
    }
    __syncthreads();
    best = red[0];
    best_i = red_i[0];
    for (int i = 1; i < nw; ++i)
      if (red[i] > best || (red[i] == best && red_i[i] < best_i)) {
        best = red[i];
        best_i = red_i[i];
      }
  }
  if (tid == 0) {
    const int prev = tok[a.cur_len - 1];
    const float s_final = text_dead ? s_ts : s_text + s_ts;
    const float logprob = (best - m_all) - logf(s_final);  // log_softmax of the filtered logits at the argmax
    int next = best_i;
    if (best_i == 0x7fffffff) next = a.eot;  // every token filtered out (cannot happen with the stock filters)
    if (prev == a.eot) next = a.eot;         // finished rows keep emitting EOT and stop accumulating
    else a.sum_logprob[b] += logprob;
    tok[a.cur_len] = next;
    if (next == a.eot) atomicAdd(a.n_done + a.cur_len, 1);
  }
}

// probs_at_sot[no_speech] of DecodingTask._main_loop (i == 0): softmax over the vocabulary of the logits at the <|sot|>
// position, probability of the <|nospeech|> token; one workgroup per row.
__global__ __launch_bounds__(1024) void token_prob_kernel(const float* __restrict__ logits, int ld, int n_vocab, int token,
                                                          float* __restrict__ out) {
  __shared__ float red[16];
  const float* lg = logits + (long)blockIdx.x * ld;
  float m = -INFINITY;
  for (int v = threadIdx.x; v < n_vocab; v += blockDim.x) m = fmaxf(m, lg[v]);
  m = block_max(m, red);
  float s = 0.f;
  for (int v = threadIdx.x; v < n_vocab; v += blockDim.x) s += __expf(lg[v] - m);
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = __expf(lg[token] - m) / s;
}

}  // namespace

hipError_t launch_token_prob(const float* logits, int ld, int n_vocab, int token, float* out, int B, hipStream_t s) {
  if (token < 0 || token >= n_vocab) return hipErrorInvalidValue;
  hipLaunchKernelGGL(token_prob_kernel, dim3(B), dim3(1024), 0, s, logits, ld, n_vocab, token, out);
  return hipGetLastError();
}

hipError_t launch_embed_step(const int* tokens, int T_max, int t, const half_t* tok_emb, const float* pos_emb, float* x, int B, int d,
                             int n_vocab, hipStream_t s) {
  hipLaunchKernelGGL(embed_step_kernel, dim3(B), dim3(256), 0, s, tokens, T_max, t, tok_emb, pos_emb, x, d, n_vocab);
  return hipGetLastError();
}

hipError_t launch_kv_append(const half_t* qkv, half_t* kc, half_t* vc, int B, int T_max, int t, int d, hipStream_t s) {
  if ((d & 7) != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kv_append_kernel, dim3(B), dim3(256), 0, s, qkv, kc, vc, T_max, t, d);
  return hipGetLastError();
}

hipError_t launch_decode_select(const DecodeSelectArgs& a, int B, hipStream_t s) {
  if (a.cur_len < 1 || a.cur_len >= a.T_max || a.n_initial < 1 || a.cur_len < a.n_initial) return hipErrorInvalidValue;
  hipLaunchKernelGGL(decode_select_kernel, dim3(B), dim3(1024), 0, s, a);
  return hipGetLastError();
}

}  // namespace wca
