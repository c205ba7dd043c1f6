// Host-side launch interface of the HIP kernels (internal to libwca.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wca {

typedef _Float16 half_t;

// ---------------------------------------------------------------- A/B and test switches (debug_switch.cpp; wca_test_set_switch)
enum { DBG_ATTN_SPLIT_VARIANT = 0, DBG_ATTN_VARIANT, DBG_HEAD_STATS_GENERAL, DBG_GEMM_SUPERTILE, DBG_LN_PAIR_V4, DBG_FAIL_PRECISION_ALLOC,
       DBG_ATTN_SPLIT_DROP, DBG_GEMM_RING, DBG_SWITCH_COUNT };
int debug_switch(int id);                            // current value (its environment variable, if any, read once as the initial value)
int set_debug_switch(const char* name, int value);   // 0, or -1 for an unknown name

// ---------------------------------------------------------------- GEMM (gemm.hip)
// C[m][n] = epilogue( sum_k A[m][k] * W[n][k] ),  A/W f16, fp32 accumulate on MFMA.
// Rows of A and C may be "batch strided": logical row m = b * rows_per_batch + t lives at
// base + b * batch_stride + t * ld  (this is how the conv stem reads overlapping windows).
struct GemmArgs {
  const half_t* A;
  long a_lo;               // > 0: A rows are [hi(K) | lo(K)] pairs, lo half a_lo elements after the hi half, and W is the PLAIN [N][K]
                           // matrix: C = (A_hi + A_lo) W^T with each W K-tile staged once (gemm256p SPLITW; only where
                           // gemm_splitw_supported() says so -- elsewhere the caller passes K-doubled operands [A_hi | A_lo], [W | W])
  int lda;                 // elements between consecutive A rows inside a batch
  int a_rows_per_batch;    // 0 => flat
  long a_batch_stride;     // elements
  const half_t* W;         // [N][ldw]
  int ldw;
  const float* bias;       // [N] or nullptr
  void* C;                 // f16 or f32, see `out_mode`
  int ldc;
  int c_rows_per_batch;    // 0 => flat
  long c_batch_stride;     // elements
  long c_lo;               // out_mode 4: elements from a value's hi half to its lo half (a multiple of 8)
  const float* addend;     // optional PRE-activation addend addend[m * ld_addend + n] (f32): added to the accumulator (with the bias) before GELU / the store --
                           // the A_hi W_lo^T term of a weight matrix that is not exact in f16 (engine.hip, W_lo slab); out_mode 0 / 1 / 4 of the tile kernels
  int ld_addend;
  const float* pos;        // optional additive table pos[(m % pos_period)][N] (f32), or nullptr
  int pos_period;
  int M, N, K;             // K % 64 == 0
  int gelu;                // apply exact (erf) GELU after bias
  int out_mode;            // 0: store f16, 1: store f32, 2: f32 accumulate (C += result),
                           // 4: store the value as an f16 PAIR hi = f16(v) at C, lo = f16(v - hi) at C + c_lo (reference-precision
                           //    "split" mode, see wca_set_precision; GELU is then the erff form, not the 1.5e-7 polynomial),
                           // 3: f32 accumulate + LayerNorm of the updated row -> ln_out (f16); the row statistics are
                           //    exchanged between the N/256 workgroups that share a 256-row panel (gemm_epilogue.h)
  unsigned a_bytes, w_bytes; // valid bytes behind A / W (buffer-descriptor bounds); 0 => derived for flat layouts
  unsigned long long* dbg; // diagnostic builds only: s_memtime stamps (never set by the product path)
  int dbg_wrap_kind;       // diagnostic builds only: bits 0-1: 0 wrap operand AND output addresses, 1 operands only, 2 outputs only; bit 2: packed (contiguous) DMA sources
  int dbg_wrap_m, dbg_wrap_n; // diagnostic builds only: tile coordinates taken modulo these (an L2-resident operand footprint; outputs collide)
  int force_tile;          // 0 auto, 128 or 256: force a tile shape (tests)
  // out_mode 3 only (the residual GEMMs of a transformer block, N = n_state <= 2048, a multiple of 256):
  const float* ln_gamma;   // [N]
  const float* ln_beta;    // [N]
  half_t* ln_out;          // [M][ln_ld] f16: LayerNorm(C) with C the updated residual row
  int ln_ld;
  float ln_eps;
  unsigned long long* ln_stats;  // workspace [N/256][ceil(M/256)*256] x {mean, M2} f32 pairs (written + read inside the launch)
  unsigned* ln_cnt;        // workspace [ceil(M/256)] arrival counters, ZEROED by launch_gemm before the launch
  int* ln_err;             // device int: bit 1 is raised if a workgroup gave up waiting for its panel's statistics
  int site;                // 0 generic, 1 encoder block (QKV / out-projection / fc1), 2 decoder, 3 conv stem / cross-KV / logits,
                           // 4 encoder fc2: selects a distinct kernel symbol per call site so rocprofv3 --stats separates them
  int supertile;           // 256x256 persistent kernel: m-panels per supertile of the tile order (0 = chosen by launch_gemm)
  int cu_limit;            // > 0: the stream this launch goes to owns only that many CUs (wca_set_cu_partition): size of the persistent grid
  // ---- few-row kernel only (gemm_rows.hip, launch_gemm_rows):
  const float* A32;        // non-null: the A operand is LayerNorm(A32 rows; ln_gamma, ln_beta, ln_eps) computed in the prologue
  int lda32;               // floats between A32 rows (K == row length)
  half_t* kv_k;            // non-null (out_mode 0, N == 3 kv_d): columns [kv_d, 2 kv_d) of row m go to kv_k + m * kv_bs + kv_t * kv_d,
  half_t* kv_v;            // columns [2 kv_d, 3 kv_d) to kv_v + ... (the step's KV-cache append); columns [0, kv_d) to C as usual
  long kv_bs;              // elements between batch rows of a cache plane (T_max * d)
  int kv_t, kv_d;
  size_t sk_bytes;         // launch_gemm (128 x 128 tile kernel, out_mode 2, few tiles and K >= 2048): bytes behind sk_part; it then
                           // splits K over up to 4 workgroups per tile ([slice][M][N] f32 partials + an ordered reduce kernel)
  float* sk_part;          // split-K workspace (gemm_rows_workspace_bytes) and
  unsigned* sk_cnt;        // one arrival counter per (64-row block, 16-column group), zero before the first launch (self-cleaning)
  int splitk;              // workgroups sharing K (0 = chosen by launch_gemm_rows: smallest with K / splitk <= 1024)
  int groups;              // 16-column groups per workgroup (0 = chosen: grid.x <= 256)
};
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);
// Few-row GEMM (M <= a few hundred rows: greedy-decode steps, batch-1 decoder forwards) with optional LayerNorm prologue,
// KV-cache append and deterministic split-K; returns hipErrorInvalidValue for shapes it does not take (gemm_rows_supported)
hipError_t launch_gemm_rows(const GemmArgs& a, hipStream_t s);
bool gemm_rows_supported(int M, int N, int K, bool layernorm_a);
int gemm_rows_pick_splitk(int K);
size_t gemm_rows_workspace_bytes(int M, int N, int splitk);
// out_mode 3 (residual + LayerNorm epilogue) is available for this shape on a device with n_cu compute units
bool gemm_ln_supported(int M, int N, int K, int n_cu);
// the pair-operand form (GemmArgs.a_lo > 0, plain W) is available for this flat-A problem (launch_gemm picks the persistent 256 x 256 kernel)
bool gemm_splitw_supported(int M, int N, int K, int lda, int out_mode);

// ---------------------------------------------------------------- attention (attention.hip)
// Flash-style multi-head attention with head_dim == 64 (every Whisper size).
// Optionally writes the pre-softmax logits q.k*scale (f32) of the first `cap_cols` keys.
struct AttnArgs {
  const half_t* Q; long q_bs; int q_rs;   // element strides: batch, row
  const half_t* K; long k_bs; int k_rs;
  const half_t* V; long v_bs; int v_rs;
  half_t* O; long o_bs; int o_rs;
  float* cap;                              // nullptr => no capture
  long cap_bs; long cap_hs; int cap_ld;    // cap[b*cap_bs + h*cap_hs + q*cap_ld + key]
  int cap_cols;                            // keys [0, cap_cols) are captured (cap_ld % 4 == 0, cap_ld >= roundup4(cap_cols))
  int nq, nk, H, B;
  float scale;                             // applied to q.k (head_dim^-0.5)
  int causal;
  unsigned long long* dbg;                 // diagnostic builds only (s_memtime stamps)
  int variant;                             // 0 auto, 1 force the 16x16x32 kernel, 2 force the 32x32x16 kernel (tests)
  // split-f16 operands (reference-precision mode): every value x is carried as hi = f16(x), lo = f16(x - hi). The pointers above
  // address the hi halves; the lo half of an element lives `*_lo` elements further (same strides). split != 0 selects
  // attn_split_kernel: S = Qhi.Khi + Qhi.Klo + Qlo.Khi, O = Phi.Vhi + Phi.Vlo + Plo.Vhi, fp32 softmax on the exact logits.
  int split;
  long q_lo, k_lo, v_lo, o_lo;
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);
hipError_t launch_attention_split(const AttnArgs& a, hipStream_t s);  // attention_split.hip (launch_attention forwards a.split != 0 here)
// (diagnostic switch DBG_ATTN_SPLIT_DROP, 0 in the product: leaves single passes out of the encoder's three-pass attention -- bit 0 K_lo Q_hi,
// bit 1 K_hi Q_lo, bit 2 V_lo P_hi, bit 3 V_hi P_lo; masks 0, 1, 2, 3, 4, 8, 9, 12, 15 are instantiated)

// ---------------------------------------------------------------- small ops (elementwise.hip)
// ld_out: elements between output rows (0 = d). lo_off != 0: split output, hi = f16(y) at out, lo = f16(y - hi) at out + lo_off
hipError_t launch_layernorm_f16(const float* x, const float* gamma, const float* beta, half_t* out,
                                int rows, int d, float eps, hipStream_t s, int ld_out = 0, long lo_off = 0);
// x[b*n + i][:] = tok_emb[tokens[b*n+i]][:] + pos_emb[i][:]
// ids outside [0, n_vocab) are embedded as token 0 and raise *err (device int, nullable)
// tok_emb_lo (nullable): the lo halves of an embedding table that is not exact in f16 (value = hi + lo)
hipError_t launch_embed(const int64_t* tokens, const half_t* tok_emb, const float* pos_emb, float* x,
                        int B, int n, int d, int n_vocab, int* err, hipStream_t s, const half_t* tok_emb_lo = nullptr);
hipError_t launch_fill_f16(half_t* p, size_t n, float v, hipStream_t s);

// ---------------------------------------------------------------- greedy ASR decode steps (decode.hip)
// x[b][:] = tok_emb[tokens[b*T_max + t]][:] + pos_emb[t][:]
hipError_t launch_embed_step(const int* tokens, int T_max, int t, const half_t* tok_emb, const float* pos_emb, float* x, int B, int d,
                             int n_vocab, hipStream_t s);
// k / v columns of qkv [B][3d] -> kc / vc [B][T_max][d] at position t
hipError_t launch_kv_append(const half_t* qkv, half_t* kc, half_t* vc, int B, int T_max, int t, int d, hipStream_t s);
// logit filters + greedy update of one decoding step (upstream decoding.py: SuppressBlank, SuppressTokens,
// ApplyTimestampRules, GreedyDecoder.update at temperature 0); one workgroup per batch row
struct DecodeSelectArgs {
  const float* logits;                 // [B] rows, ld apart: the last position's fp32 logits
  int ld, n_vocab;
  int* tokens;                         // [B][T_max]; row b holds cur_len tokens, the new one is written at [cur_len]
  int T_max, cur_len, n_initial;
  const unsigned char* suppress_mask;  // [n_vocab] 1 = always -inf (SuppressTokens, <|notimestamps|>)
  const unsigned char* blank_mask;     // [n_vocab] 1 = -inf on the first sampled position (SuppressBlank); nullable
  int eot, timestamp_begin;
  int apply_timestamp_rules, max_initial_timestamp_index;  // index < 0: no limit
  float* sum_logprob;                  // [B] accumulated log-probability of the sampled tokens
  int* n_done;                         // [T_max] n_done[cur_len] += 1 for every row whose new token is EOT
};
hipError_t launch_decode_select(const DecodeSelectArgs& a, int B, hipStream_t s);
// out[b] = softmax(logits[b])[token]  (no_speech_prob: the <|nospeech|> probability at the <|sot|> position)
hipError_t launch_token_prob(const float* logits, int ld, int n_vocab, int token, float* out, int B, hipStream_t s);
hipError_t launch_f32_to_f16(const float* in, half_t* out, size_t n, hipStream_t s);

// ---------------------------------------------------------------- log-mel (logmel.hip)
struct LogMelArgs {
  const float* pcm;        // [B][pcm_stride] f32 16 kHz, already trimmed/padded by the caller or shorter
  long pcm_stride;
  const int* n_samples;    // [B] valid samples per utterance (<= 480000); device pointer
  const float* filters;    // [n_mels][201]
  const int* filt_lo;      // [n_mels] first non-zero bin of each filter (device)
  const int* filt_hi;      // [n_mels] one past the last non-zero bin (device)
  const float* window;     // [400] periodic hann
  const float* twiddle;    // [400][2] cos,sin(2*pi*m/400)
  float* mel_out;          // [B][n_mels][3000] f32 (may be nullptr)
  half_t* mel_tm;          // [B][3002][n_mels_pad] f16 time-major, rows 0 and 3001 zero (may be nullptr)
  int n_mels_pad;          // row length of mel_tm
  int tm_lo;               // != 0: mel_tm carries split values, hi at column m, lo = f16(v - hi) at column tm_lo + m
  int precise;             // != 0: the DFT and the filterbank sums accumulate in f64 (reference-precision mode)
  float* scratch;          // [B][n_mels][3000] raw log10 values (required)
  unsigned* gmax;          // [B] ordered-int encoded running max (required)
  int n_mels, B;
};
hipError_t launch_logmel(const LogMelArgs& a, hipStream_t s);

// ---------------------------------------------------------------- post-processing (postproc.hip)
struct HeadStatsArgs {
  const float* qk;         // captured logits: qk[b*qk_bs + head*qk_hs + t*qk_ld + f]
  long qk_bs; long qk_hs; int qk_ld;
  float* weights;          // optional dense out: weights[b*w_bs + head*n*F... ] see w_* (nullptr => skip)
  long w_bs;               // elements between utterances
  const int* n_tok;        // [B] decoder rows per utterance (device)
  const int* n_frames;     // [B] F per utterance (device)
  int n_tok_max, n_frames_max;  // strides of the dense `weights` layout: [head][n_tok_max][n_frames_max]
  float* colnorm;          // [B][LH][n_frames_max] per-head column L2 norms
  float* scores;           // [B][LH]
  float* rowstats;         // optional [B][LH][n_tok_max][2] = (row max of med*scale, sum of exp) so that selected
                           // heads can be re-materialised later without storing `weights`
  int LH, B, medfilt_width;
  float qk_scale, w_col, w_row, w_cov;
  int input_is_weights;    // 1: `qk` already holds softmaxed weights (filter_attention on a given tensor): no median/softmax
};
hipError_t launch_head_stats(const HeadStatsArgs& a, hipStream_t s);

// ascending top-k (python tuple order: score, then flat head index); k_eff = min(k, LH)
hipError_t launch_topk(const float* scores, int LH, int B, int k, int* sel_idx /*[B][k]*/,
                       float* sel_score /*[B][k]*/, hipStream_t s);

struct AggregateArgs {
  const float* weights; long w_bs; int n_tok_max, n_frames_max;
  const float* colnorm;    // [B][LH][n_frames_max]
  const int* sel_idx;      // [B][n_sel] head ids (or nullptr => heads [head_lo, LH))
  int n_sel, head_lo, LH, B;
  const int* n_tok; const int* n_frames;
  int row_lo, row_hi_trim; // output rows [row_lo, n_tok - row_hi_trim)
  float* matrix;           // [B][n_tok_max][n_frames_max]
  // recompute mode (weights == nullptr): re-derive the softmaxed value from the captured logits + rowstats
  const float* qk; long qk_bs; long qk_hs; int qk_ld;
  const float* rowstats;   // [B][LH][n_tok_max][2]
  int medfilt_width; float qk_scale;
};
hipError_t launch_aggregate(const AggregateArgs& a, hipStream_t s);

// default_find_alignment (timing.py:159-163): out[s][t][f] = (ws[sel[s]][t][f] - mean_t) / std_t (population std over
// the n token rows), then matrix = mean over s of rows [row_lo, n - row_hi_trim). ws [LH][n][F], out [n_sel][n][F].
hipError_t launch_stdmean_normalize(const float* ws, const int* sel_dev, int n_sel, int n, int F, float* out, hipStream_t s);
hipError_t launch_mean_heads(const float* x, int n_sel, int n, int F, int row_lo, int row_hi_trim, float* matrix, hipStream_t s);

// per-head strict word-boundary scoring of the probe (metrics.py:45-72 over every head of probe_oracle.py:83-90): tp_out[hd]
hipError_t launch_probe_strict(const int* jump, int jump_ld, int LH, const int* wb_end, int n_hyp, const double* y, int n_ref,
                               const unsigned char* eq, double tol, int* tp_out, hipStream_t s);

// standalone median filter along the last axis with reflect padding (whisper.timing.median_filter)
hipError_t launch_median_filter(const float* in, float* out, long rows, int F, int width, hipStream_t s);

// ---------------------------------------------------------------- DTW (dtw.hip)
struct DtwArgs {
  const float* matrix;     // P problems: matrix + p*m_bs, rows ld apart; DTW runs on the NEGATED matrix
  long m_bs; int ld;
  const int* N;            // [P] rows per problem (device)   (or nullptr => N_all)
  const int* M;            // [P] cols per problem (device)   (or nullptr => M_all)
  int N_all, M_all;
  int N_max, M_max;
  uint32_t* trace;         // workspace: P * N_max * ceil(M_max/16) words
  int* path;               // [P][2][cap] text idx row then time idx row, right-aligned; cap = N_max + M_max + 2
  int* path_len;           // [P]
  int* jump_frame;         // [P][jump_ld] first frame of each text row (or nullptr)
  int jump_ld;             // row stride of jump_frame (>= N_max)
  int P;
};
hipError_t launch_dtw(const DtwArgs& a, hipStream_t s);

}  // namespace wca
