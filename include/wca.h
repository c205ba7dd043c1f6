/*
 * wca.h -- C ABI of libwca.so, the MI355X (gfx950) forced-alignment engine.
 *
 * This is the drop-in boundary for the hot path of 30stomercury/whisper-char-alignment.
 * The reference has no FFI layer of its own (it is pure Python calling openai-whisper); the
 * boundary is therefore the set of Python functions below, and each entry point here names the
 * reference interface it replaces (file:line in the reference repository):
 *
 *   wca_log_mel            dataset.py:46-48, dataset.py:107-109, README.md:101-103
 *                          (whisper.pad_or_trim + whisper.log_mel_spectrogram)
 *   wca_get_attentions     timing.py:45-67   get_attentions(): teacher-forced forward with every
 *                          cross-attention QK captured (timing.py:50-58), [:max_frames] slice,
 *                          median_filter, *qk_scale, softmax (timing.py:63-66); logits returned
 *   wca_attention_weights  timing.py:63-66   the same post-processing on caller-supplied captured logits
 *   wca_median_filter      timing.py:65      whisper.timing.median_filter
 *   wca_filter_attention   timing.py:13-43   filter_attention(): head scores + tuple-ordered top-k
 *                          (+ metrics.py:99-111 coverage_penalty)
 *   wca_force_align        timing.py:69-103  force_align() up to and including dtw(-matrix):
 *                          aggregation "mean" (timing.py:84-89) / "topk" (timing.py:91-97), the
 *                          [len(sot_sequence):-1] slice (timing.py:102), DTW + backtrace (timing.py:103)
 *   wca_default_find_alignment  timing.py:116-186 default_find_alignment (std/mean normalised alignment heads)
 *   wca_dtw                timing.py:103     whisper.timing.dtw -> dtw_cpu + backtrace
 *   wca_probe_heads        probe_oracle.py:83-90  per-head force_align sweep (one DTW per head)
 *   wca_greedy_decode      infer_ali.py:40,60-61, probe_oracle.py:59-60, README.md:107-108:
 *                          whisper.decode(model, mels, DecodingOptions(language="en")) -- the greedy ASR pre-pass
 *                          that produces the teacher text (upstream decoding.py: KV-cached autoregressive
 *                          decoder, SuppressBlank / SuppressTokens / ApplyTimestampRules, GreedyDecoder)
 *   wca_align_batch        infer_ali.py:93-101 + dataset.py:47-48: the whole per-utterance pipeline
 *                          (log-mel -> forward+capture -> medfilt/softmax -> scores/top-k ->
 *                          aggregate -> DTW) for a micro-batch of utterances, results = the frame
 *                          at which the DTW path enters every token row (timing.py:110-111 `jumps`)
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (streams are passed as void*).
 *   - `_dev` pointers are device (HBM) pointers owned by the caller; `_host` pointers are host memory.
 *   - every function returns 0 on success or a negative wca_status; wca_last_error() gives a
 *     thread-local description. Nothing throws across this boundary.
 *   - one engine <-> one GPU <-> one stream <-> one host thread. Engines are independent.
 *   - limits mirror infer_ali.py:25-26,79: n_tok <= 448, max_frames <= 1500 (WCA_ERR_TOO_LONG).
 */
#ifndef WCA_H_
#define WCA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wca_engine wca_engine;

typedef enum {
  WCA_STATUS_PARTITIONED = 1, /* wca_engine_set_stream on a CU-partitioned engine: the stream was RECORDED (it becomes the engine's stream
                               * again when the partition is lifted) but phase 1 stays on its CU-masked stream, which is not ordered with
                               * the caller's: synchronise before and after each entry point (the Python binding does) */
  WCA_OK = 0,
  WCA_ERR_INVALID = -1,   /* bad argument / shape                                  */
  WCA_ERR_TOO_LONG = -2,  /* n_tok > 448 or max_frames > 1500 (infer_ali.py:79)    */
  WCA_ERR_HIP = -3,       /* a HIP runtime call failed                             */
  WCA_ERR_STATE = -4,     /* weights missing / engine not finalized                */
  WCA_ERR_NOMEM = -5
} wca_status;

/* whisper.model.ModelDimensions */
typedef struct {
  int32_t n_mels;
  int32_t n_audio_ctx;    /* 1500 */
  int32_t n_audio_state;
  int32_t n_audio_head;
  int32_t n_audio_layer;
  int32_t n_vocab;
  int32_t n_text_ctx;     /* 448 */
  int32_t n_text_state;
  int32_t n_text_head;
  int32_t n_text_layer;
} wca_model_dims;

enum { WCA_DTYPE_F32 = 0, WCA_DTYPE_F16 = 1 };
enum { WCA_AGGR_MEAN = 0, WCA_AGGR_TOPK = 1 };

/* options of filter_attention / force_align (timing.py:13, timing.py:69-78) */
typedef struct {
  int32_t aggregation;    /* WCA_AGGR_MEAN | WCA_AGGR_TOPK                          */
  int32_t topk;           /* > 0 required for WCA_AGGR_TOPK (timing.py:92)          */
  float w_colnorm;        /* default 1.0                                           */
  float w_rownorm;        /* default 1.0                                           */
  float w_coverage;       /* default 0.0                                           */
  int32_t sot_len;        /* len(tokenizer.sot_sequence): rows dropped at the top   */
  int32_t medfilt_width;  /* odd; used by wca_align_batch / wca_get_attentions      */
  float qk_scale;         /* 1.0 in the reference (infer_ali.py:45)                 */
} wca_align_opts;

const char* wca_last_error(void);
int wca_version(void);

/* ---- engine lifetime ------------------------------------------------------------------------ */
/* A new engine is in the CONTRACT precision mode (WCA_PRECISION_REFERENCE: every stage on (hi, lo) operand pairs = the fp32 forward of
 * /root/reference/timing.py:58 to fp32 summation noise; see wca_set_precision). Since round 5 the fast f16-operand mode is the opt-in:
 * wca_engine_create_ex(..., WCA_PRECISION_F16, ...) builds the engine directly in it (no wide arena is ever allocated), or switch later
 * with wca_set_precision. */
int wca_engine_create(const wca_model_dims* dims, int device_ordinal, int max_batch, wca_engine** out);
int wca_engine_create_ex(const wca_model_dims* dims, int device_ordinal, int max_batch, int precision_mode, wca_engine** out);
void wca_engine_destroy(wca_engine* e);
/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the HIP default (null)
 * stream. Until this is called the engine runs on a private non-blocking stream of its own. */
int wca_engine_set_stream(wca_engine* e, void* hip_stream);
int wca_engine_synchronize(wca_engine* e);

/* Weights in openai-whisper state_dict naming ("encoder.blocks.0.attn.query.weight", ...), host
 * memory, row-major. Also accepts the auxiliary table "mel_filters" [n_mels][201] (whisper's
 * assets/mel_filters.npz). Call wca_finalize_weights once after the last tensor. */
int wca_load_weight(wca_engine* e, const char* name, const void* host_ptr, int dtype, const int64_t* shape, int ndim);
int wca_finalize_weights(wca_engine* e);
/* Weight MATRICES (Linear / Conv1d weights, the token embedding) are stored f16, like every openai checkpoint at rest
 * (/root/reference/infer_ali.py:36-37: whisper.load_model upcasts those f16 values to fp32 parameters); biases, LayerNorm parameters and
 * positional embeddings stay fp32. An fp32 source tensor whose values are NOT f16-representable (a fine-tuned fp32 state dict, which the
 * reference runs in true fp32) is not rounded away: wca_load_weight keeps the remainder lo = f16(w - f16(w)) of every such element in a second
 * slab (w = hi + lo to 2^-22 |w|, the representation the activations travel in) and counts the elements per tensor (wca_weights_inexact:
 * tensors, values, name of the first one). While a precision site is on pairs its GEMMs multiply the extra term A_hi W_lo^T for those
 * matrices (one more f16 pass: an accumulating launch for the residual GEMMs, a pre-activation addend for the others; the embedding adds
 * hi + lo) -- slower, never a narrower model. wca_set_allow_rounded_weights(e, 1) drops the remainders (the engine then computes what the
 * f16-rounded checkpoint computes); the f16 mode (approximate by definition) ignores them. */
int wca_weights_inexact(wca_engine* e, long long* n_tensors_out, long long* n_values_out, char* first_name_out, int first_name_cap);
int wca_set_allow_rounded_weights(wca_engine* e, int on);

/* ---- hot path, one reference function per entry point ---------------------------------------- */

/* pcm_dev: [batch][pcm_stride] f32 (values beyond n_samples are ignored = zero padding to 30 s);
 * mel_out_dev: [batch][n_mels][3000] f32. */
int wca_log_mel(wca_engine* e, const float* pcm_dev, int64_t pcm_stride, const int32_t* n_samples_host, int batch,
                float* mel_out_dev);

/* mel_dev [batch][n_mels][3000] f32; tokens_dev [batch][n_tok] int64 (already sot..eot framed,
 * infer_ali.py:69-76; shorter utterances padded with any valid token id, true lengths in n_tok_host,
 * NULL = all n_tok); max_frames_host [batch].
 * weights_out_dev: [batch][L][H][n_tok][F] f32 with F = max over the batch of max_frames (rows/cols
 * beyond an utterance's own n_tok/max_frames are unspecified); logits_out_dev: [batch][n_tok][n_vocab]
 * f32 or NULL. */
int wca_get_attentions(wca_engine* e, const float* mel_dev, const int64_t* tokens_dev, int batch, int n_tok,
                       const int32_t* n_tok_host, const int32_t* max_frames_host, int medfilt_width, float qk_scale,
                       float* weights_out_dev, float* logits_out_dev);

/* The post-capture half of get_attentions (timing.py:63-66) on caller-supplied logits: qk_dev [L][H][n][ld] f32 (what
 * the forward hooks collected, torch.cat'ed over layers; ld = row stride >= max_frames, 1500 in the reference) ->
 * [..., :max_frames] slice, median_filter(medfilt_width), * qk_scale, softmax over frames ->
 * weights_out_dev [L][H][n][max_frames] f32. */
int wca_attention_weights(wca_engine* e, const float* qk_dev, int L, int H, int n, int ld, int max_frames,
                          int medfilt_width, float qk_scale, float* weights_out_dev);

/* in/out [rows][F] f32 device, reflect padding, width odd */
int wca_median_filter(wca_engine* e, const float* in_dev, float* out_dev, int64_t rows, int F, int width);

/* attns_dev [L][H][n][F] f32 (already softmaxed). scores_host [L*H]; sel_idx_host/sel_score_host
 * [min(topk, L*H)] ascending by (score, (l,h)) -- element i is head l = idx / H, h = idx % H. */
int wca_filter_attention(wca_engine* e, const float* attns_dev, int L, int H, int n, int F, int topk, float w_colnorm,
                         float w_rownorm, float w_coverage, float* scores_host, int32_t* sel_idx_host,
                         float* sel_score_host);

/* ws_dev [L][H][n][F] f32. Outputs (host): matrix_host [(n - sot_len - 1)][F] f32 (the sliced,
 * NOT negated matrix, timing.py:102); text_idx_host / time_idx_host [n - sot_len - 1 + F] int32 with
 * *path_len_host valid entries; sel_idx_host / sel_score_host [topk] (topk mode only, may be NULL). */
int wca_force_align(wca_engine* e, const float* ws_dev, int L, int H, int n, int F, const wca_align_opts* opts,
                    float* matrix_host, int32_t* text_idx_host, int32_t* time_idx_host, int32_t* path_len_host,
                    int32_t* sel_idx_host, float* sel_score_host);

/* timing.py:116-186 default_find_alignment (openai-whisper's own aligner, `--default_whisper_timing`), the part after
 * median filter + softmax: weights of the given alignment heads (flat ids l*H + h) are normalised per head and frame
 * over the token axis, (w - mean) / std with population std (timing.py:159-160), averaged over the heads (:162),
 * sliced [sot_len:-1] (:163) and aligned with dtw(-matrix) (:165; the dtw_cpu tie rule is used).
 * ws_dev [L][H][n][F] as returned by wca_get_attentions; heads_host in the order of model.alignment_heads.indices().T
 * (row-major over (l, h)); weights_norm_out_dev [n_heads][n][F] f32 device or NULL: the normalised weights, which
 * the reference returns as its 4th value (timing.py:186); other outputs as in wca_force_align. */
int wca_default_find_alignment(wca_engine* e, const float* ws_dev, int L, int H, int n, int F, const int32_t* heads_host,
                                int n_heads, int sot_len, float* weights_norm_out_dev, float* matrix_host,
                                int32_t* text_idx_host, int32_t* time_idx_host, int32_t* path_len_host);

/* DTW on the NEGATED matrix exactly like `dtw(-matrix)`; matrix_host [N][M] f32 row-major.
 * text_idx_host / time_idx_host capacity N + M. */
int wca_dtw(wca_engine* e, const float* matrix_host, int N, int M, int32_t* text_idx_host, int32_t* time_idx_host,
            int32_t* path_len_host);

/* Same for P independent problems already in HBM (probe_oracle.py:88-90: one DTW per head).
 * matrix_dev [P][N][M]; jump_frame_host [P][N]: frame at which the path enters each row. */
int wca_dtw_batch_dev(wca_engine* e, const float* matrix_dev, int P, int N, int M, int32_t* jump_frame_host);

/* probe_oracle.py:83-90: one alignment PER HEAD (aggregation "mean" on a single head = column
 * normalisation only), all L*H DTWs in one launch. ws_dev [L][H][n][F]; scores_host [L*H] = the
 * filter_attention scores (w_col = w_row = 1); jump_frame_host [L*H][n - sot_len - 1]. */
int wca_probe_heads(wca_engine* e, const float* ws_dev, int L, int H, int n, int F, int sot_len, float* scores_host,
                    int32_t* jump_frame_host);
/* Strict word-boundary scoring of EVERY head of the preceding wca_probe_heads (probe_oracle.py:83-90 calling
 * metrics.py:45-72 eval_n1_strict once per head): head hd's predicted end of hypothesis word i is
 * jump_frame[hd][word_end_row[i]] / 50 s; it is a true positive if an unused reference boundary j with
 * same_word[i * n_ref + j] != 0 lies within `tolerance` (first such j, the reference's loop order). tp_host [n_heads = L*H];
 * fp = n_hyp - tp, fn = n_ref - tp. Times are compared in float64 like the host code: tp is the same integer. */
int wca_probe_strict_tp(wca_engine* e, int n_heads, const int32_t* word_end_row_host, int n_hyp, const double* ref_times_host, int n_ref,
                        const uint8_t* same_word_host, double tolerance, int32_t* tp_host);

/* Fused per-utterance pipeline for a micro-batch (the north-star hot path).
 * pcm_dev [batch][pcm_stride] f32 (NULL: re-use the encoder state of the preceding wca_greedy_decode of this
 * batch); tokens_dev [batch][n_tok_max] int64; n_tok_host, n_samples_host, max_frames_host [batch].
 * jump_frame_host [batch][n_tok_max]: for utterance b, entries [0, n_tok[b] - sot_len - 1) are the frame
 * index at which the DTW path enters that text row (jump_times * 50, timing.py:110-111); the remaining entries of a row are 0.
 * sel_idx_host [batch][topk] may be NULL. */
int wca_align_batch(wca_engine* e, const float* pcm_dev, int64_t pcm_stride, const int32_t* n_samples_host,
                    const int64_t* tokens_dev, int n_tok_max, const int32_t* n_tok_host,
                    const int32_t* max_frames_host, int batch, const wca_align_opts* opts, int32_t* jump_frame_host,
                    int32_t* sel_idx_host);

/* whisper.DecodingOptions as the reference uses it (infer_ali.py:40: language="en", everything else default:
 * task transcribe, temperature 0 -> greedy, no beam, sample_len n_text_ctx // 2, suppress_blank, suppress_tokens "-1",
 * without_timestamps False, max_initial_timestamp 1.0, no prompt / prefix). */
typedef struct {
  int32_t sample_len;                  /* maximum number of sampled tokens (224)                              */
  int32_t eot;                         /* tokenizer.eot                                                       */
  int32_t timestamp_begin;             /* tokenizer.timestamp_begin (<|0.00|>)                                */
  int32_t apply_timestamp_rules;       /* 1 = ApplyTimestampRules (without_timestamps False)                  */
  int32_t max_initial_timestamp_index; /* round(max_initial_timestamp / 0.02) = 50; < 0 = no limit            */
  int32_t no_speech;                   /* tokenizer.no_speech (<|nospeech|>) for no_speech_prob; < 0 = skip   */
} wca_decode_opts;

/* Greedy ASR pre-pass for a micro-batch. At most one of mel_dev ([batch][n_mels][3000] f32, what whisper.decode
 * takes) and pcm_dev ([batch][pcm_stride] f32 + n_samples_host, log-mel computed on the device) is non-NULL; with
 * both NULL the oldest undecoded state of wca_encode_batch is decoded.
 * initial_tokens_host [n_initial]: tokenizer.sot_sequence (every row starts with it).
 * suppress_mask_host [n_vocab] bytes: 1 = logit forced to -inf at every step (SuppressTokens list and, when the
 *   timestamp rules are on, <|notimestamps|>); blank_mask_host [n_vocab] (nullable): 1 = -inf at the first sampled
 *   position (SuppressBlank: tokenizer.encode(" ") + [eot]).
 * tokens_out_host [batch][n_initial + sample_len] int32 (positions never reached hold eot);
 * n_tokens_host [batch]: row b's sampled tokens before its first EOT are tokens_out[b][n_initial : n_tokens[b]];
 * sum_logprob_host [batch] (nullable): GreedyDecoder's sum of log-probabilities of the sampled tokens.
 * no_speech_prob_host [batch] (nullable; needs opts->no_speech >= 0): softmax probability of <|nospeech|> at the
 *   <|startoftranscript|> position (DecodingResult.no_speech_prob).
 * The encoder output and cross-attention K/V of this batch stay in the engine: the next wca_align_batch_enqueue
 * for the same batch may pass pcm_dev = NULL to re-use them (the reference runs the encoder twice,
 * infer_ali.py:60 and timing.py:58). Synchronous (returns with the tokens). A state that was decoded but not aligned
 * is dropped when the next wca_greedy_decode starts. */
/* Phase 1 alone (log-mel or a given mel -> encoder -> cross-attention K/V of every decoder layer), enqueued on the
 * engine stream without a host sync. The encoded state is queued: wca_greedy_decode(mel_dev = pcm_dev = NULL) decodes
 * the oldest undecoded one, wca_align_batch_enqueue(pcm_dev = NULL) consumes the oldest one. Two K/V slots exist, so
 * the NEXT batch can be encoded while the current one is decoded and aligned (the decode loop and the alignment's
 * phase 2 run on the engine's second stream). */
int wca_encode_batch(wca_engine* e, const float* mel_dev, const float* pcm_dev, int64_t pcm_stride,
                     const int32_t* n_samples_host, int batch);

int wca_greedy_decode(wca_engine* e, const float* mel_dev, const float* pcm_dev, int64_t pcm_stride,
                      const int32_t* n_samples_host, int batch, const int32_t* initial_tokens_host, int n_initial,
                      const uint8_t* suppress_mask_host, const uint8_t* blank_mask_host, const wca_decode_opts* opts,
                      int32_t* tokens_out_host, int32_t* n_tokens_host, float* sum_logprob_host,
                      float* no_speech_prob_host);

/* Same pipeline, but only enqueues the work on the engine stream (no host sync; results stay in the engine's
 * pinned staging ring until wca_align_batch_fetch). Up to TWO batches may be in flight: _fetch returns the
 * OLDEST pending one (waiting on its HIP event only), so the host can post-process batch i while batch i+1 runs. */
int wca_align_batch_enqueue(wca_engine* e, const float* pcm_dev, int64_t pcm_stride, const int32_t* n_samples_host,
                            const int64_t* tokens_dev, int n_tok_max, const int32_t* n_tok_host,
                            const int32_t* max_frames_host, int batch, const wca_align_opts* opts);
int wca_align_batch_fetch(wca_engine* e, int batch, int n_tok_max, int topk, int32_t* jump_frame_host,
                          int32_t* sel_idx_host);

/* ---- audio I/O on the host (no GPU work, no engine): dataset.py:31,104 read the corpora through torchaudio.load; the
 * LibriSpeech originals are FLAC. buf = the whole .flac file in memory.
 * wca_flac_decode: out [channels][capacity_per_channel] f32 in [-1, 1) (sample / 2^(bits-1), torchaudio's
 * normalisation); out == NULL only counts (*n_decoded = samples per channel: streams whose STREAMINFO has no total).
 * CRC-8 / CRC-16 of every frame are verified; WCA_ERR_INVALID on a malformed / unsupported stream. */
int wca_flac_info(const uint8_t* buf, int64_t nbytes, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample,
                  int64_t* total_samples);
int wca_flac_decode(const uint8_t* buf, int64_t nbytes, float* out, int64_t capacity_per_channel, int64_t* n_decoded);

/* ---- collation of the per-rank results over RCCL -----------------------------------------------------
 * The reference is single-process: infer_ali.py:53-56,118-132 accumulates every utterance's result in one dict and three
 * counters. Here utterances shard over one process per GPU; at the end every rank hands its packed records
 * ([utt_index:int32][n:int32][starts:f64 x n][ends:f64 x n] ..., shard.pack_results) to wca_allgather_results: an all-gather
 * of the byte counts, then ONE all-gather of the buffers padded to the largest count (ncclAllGather over xGMI), and
 * wca_allreduce_counters sums the evaluation counters. librccl is resolved at first use (dlopen; the copy already in the
 * process if there is one). The 128-byte unique id is created on rank 0 and reaches the other ranks by the launcher's own
 * side channel (a file, MPI, the torch.distributed store ...). One communicator per engine, on the engine's device and
 * stream. The Python CLI / bench.py collate through torch.distributed (backend "nccl" = the same RCCL) by default;
 * shard.allgather_results(..., engine=model) takes this path. */
#define WCA_COMM_ID_BYTES 128
int wca_comm_unique_id(uint8_t* id_out /* [WCA_COMM_ID_BYTES] */);
int wca_comm_init(wca_engine* e, const uint8_t* id /* [WCA_COMM_ID_BYTES] */, int rank, int world);
int wca_comm_destroy(wca_engine* e);
/* gathered_host [world][capacity_per_rank]: rank r's sizes_host[r] bytes start at r * capacity_per_rank. The first collective
 * gathers every rank's {n_bytes, capacity_per_rank}; WCA_ERR_TOO_LONG is returned ON EVERY RANK ALIKE when the largest n_bytes
 * exceeds the SMALLEST capacity any rank passed (the decision is wca_collate_plan on the gathered pairs, so all ranks issue the
 * same sequence of collectives whatever their own capacity); sizes_host is filled: retry with max(sizes_host) everywhere. */
int wca_allgather_results(wca_engine* e, const uint8_t* packed_host, int64_t n_bytes, uint8_t* gathered_host,
                          int64_t capacity_per_rank, int64_t* sizes_host /* [world] */);
int wca_allreduce_counters(wca_engine* e, int64_t* counters_host /* [n], summed in place */, int n);
/* The fit decision of wca_allgather_results as a pure host function (no GPU, no communicator): WCA_OK and *pad_out = the padded
 * per-rank payload when max(sizes) <= min(capacities), else WCA_ERR_TOO_LONG (pad_out still set). Exported for the protocol tests. */
int wca_collate_plan(const int64_t* sizes /* [world] */, const int64_t* capacities /* [world] */, int world, int64_t* pad_out);

/* ---- kernel-level entry points (used by the parity tests and by bench.py's roofline leg) -------- */
/* C[m][n] = sum_k A[m][k] W[n][k] (+bias) ; A,W f16 device. out_mode low byte: 0 f16 store, 1 f32 store,
 * 2 f32 accumulate, 4 f16 PAIR store (split mode: c_dev is f16 [M][2N], hi = f16(v) at column n, lo = f16(v - hi) at column
 * N + n; GELU is then the erff form); out_mode >> 8: force the tile shape (0 auto, 128, 256). A split-mode GEMM is this call
 * with A = [A_hi | A_lo] ([M][2K]), W = [W | W] ([N][2K]) and K = 2K. */
int wca_test_gemm(wca_engine* e, const void* a_f16_dev, const void* w_f16_dev, const float* bias_dev, void* c_dev, int M,
                  int N, int K, int gelu, int out_mode);
/* The pair product with the W K-tiles staged ONCE (persistent 256 x 256 kernel, SPLITW form): a2 = [A_hi | A_lo] ([M][2K] f16),
 * w = the PLAIN [N][K] matrix, K the algorithmic depth. out_mode as above (0, 1, 2, 4; >> 8: 0 auto, 257, 258).
 * WCA_ERR_INVALID where that kernel does not apply (fewer than 192 256 x 256 tiles, K % 128 != 0): the engine then multiplies
 * the K-doubled operands through wca_test_gemm's path. */
int wca_test_gemm_pairs(wca_engine* e, const void* a2_f16_dev, const void* w_f16_dev, const float* bias_dev, void* c_dev, int M,
                        int N, int K, int gelu, int out_mode);
/* x (f32 [M][N], read-modify-write) += A W^T + bias; xn (f16 [M][N]) = LayerNorm(x; gamma, beta, eps 1e-5): the residual
 * GEMMs of an encoder block with the LayerNorm in their epilogue (gemm_epilogue.h). site 1 / 4 = the out-projection's /
 * fc2's kernel symbol. WCA_ERR_INVALID where the fused form does not apply (N % 256, K % 128, fewer than 192 tiles). */
int wca_test_gemm_ln(wca_engine* e, const void* a_f16_dev, const void* w_f16_dev, const float* bias_dev, float* x_dev,
                     const float* gamma_dev, const float* beta_dev, void* xn_f16_dev, int M, int N, int K, int site);
/* The few-row GEMM of the greedy-decode steps (gemm_rows.hip): C [M][N] = epilogue(A W^T + bias) with A = a_f16_dev [M][K] or,
 * when x_f32_dev != NULL, A = LayerNorm(x_f32_dev [M][K]; gamma, beta, eps 1e-5) computed in the kernel's prologue.
 * out_mode as wca_test_gemm; splitk 0 = smallest split with K / splitk <= 1024, groups 0 = chosen; kv_k / kv_v != NULL
 * (out_mode 0, N = 3 d): columns [d, 3d) go to the caches [M][T_max][d] at position kv_t instead of C. */
int wca_test_gemm_rows(wca_engine* e, const void* a_f16_dev, const float* x_f32_dev, const float* gamma_dev, const float* beta_dev,
                       const void* w_f16_dev, const float* bias_dev, void* c_dev, int M, int N, int K, int gelu, int out_mode, int splitk,
                       int groups, void* kv_k_f16_dev, void* kv_v_f16_dev, int T_max, int kv_t);
/* diagnostic build of the pipelined 256x256 GEMM that records s_memtime stamps per K tile into dbg_dev
 * ([4 blocks][8 waves][64 tiles][8] u64); development aid for tools/gemm_stamps.py, never used by the product.
 * out_mode: bits 0-7 as wca_test_gemm (0 / 2 / 4), bit 8 GELU, bit 9 pair operands (a = [M][hi(K) | lo(K)], plain w: the SPLITW form),
 * bits 12-15 / 16-19: when non-zero, tile coordinates are taken modulo these (m, n) -- an L2-resident operand footprint, outputs
 * collide; with the wrap set dbg_dev may be NULL (no stamps: plain timing of the wrapped launch); bits 20-21: 0 wrap operand and output
 * addresses, 1 operand addresses only, 2 output addresses only; bit 22: every LDS-DMA piece reads 1 KiB of contiguous memory (timing only) */
int wca_test_gemm_stamped(wca_engine* e, const void* a_f16_dev, const void* w_f16_dev, void* c_dev, int M, int N, int K,
                          int out_mode, unsigned long long* dbg_dev);
/* diagnostic, process-wide (the product never calls it; 0 = the contract's three passes per product): leave single MFMA passes out of the
 * encoder's pair attention -- bit 0 K_lo Q_hi, bit 1 K_hi Q_lo, bit 2 V_lo P_hi, bit 3 V_hi P_lo; masks 0, 1, 2, 3, 4, 8, 9, 12, 15 exist.
 * tools/precision_ablation.py --attn-drop: the product-level ablation of timing.py:58's fp32 attention. */
int wca_test_set_attn_split_drop(int mask);
/* A/B and test switches of the library, process-wide (csrc/debug_switch.cpp; every default is the shipped choice and the product never calls
 * this): "attn_split_variant" (1: pair attention on the 16x16x32 kernel everywhere), "attn_variant" (f16 attention: 1 / 3), "head_stats_general"
 * (1: the general head-statistics kernel), "gemm_supertile" (m-panels per supertile), "ln_pair_v4", "fail_precision_alloc" (1: the next
 * precision switch fails its allocation: the roll-back test), "attn_split_drop", "gemm_ring" (1: the pair GEMM on round 4's two-slot rings). The environment variable of the same
 * meaning (WCA_ATTN_SPLIT_VARIANT, ...) is read ONCE, as the switch's initial value, never per launch. */
int wca_test_set_switch(const char* name, int value);
/* the [batch][n_text_layer * n_text_head] head selection scores (timing.py:13-43) of the LAST fused batch, after it was fetched (no batch in
 * flight): tools/precision_ablation.py compares them with the oracle's */
int wca_test_last_scores(wca_engine* e, int batch, float* scores_host);
/* q,k,v [B][n][H*64] f16 device -> o [B][nq][H*64] f16; cap_dev [B][H][nq][cap_ld] f32 or NULL.
 * causal: bit 0 = causal mask; bits 8-9 = kernel variant (0 auto, 1 the 16x16x32-MFMA kernel, 2 the 32x32x16-MFMA
 * kernel that serves the encoder's un-masked self-attention = what auto picks, 3 the same with the row sums on the vector
 * ALU: an experiment that was not adopted) */
int wca_test_attention(wca_engine* e, const void* q_dev, const void* k_dev, const void* v_dev, void* o_dev,
                       float* cap_dev, int cap_ld, int cap_cols, int B, int H, int nq, int nk, int causal);
/* split-mode attention (attention_split.hip): q2,k2,v2 [B][n][2*H*64] f16 rows [hi(H*64) | lo(H*64)] -> o2 [B][nq][2*H*64]
 * likewise; cap_dev as above (the three-pass fp32 logits times scale). causal: bit 0 only. */
int wca_test_attention_split(wca_engine* e, const void* q2_dev, const void* k2_dev, const void* v2_dev, void* o2_dev,
                             float* cap_dev, int cap_ld, int cap_cols, int B, int H, int nq, int nk, int causal);
/* diagnostic build with s_memtime stamps per key tile ([4 blocks][4 waves][32 tiles][8] u64); tools/attn_stamps.py */
int wca_test_attention_stamped(wca_engine* e, const void* q_dev, const void* k_dev, const void* v_dev, void* o_dev, int B, int H,
                               int nq, int nk, unsigned long long* dbg_dev);
/* one step of the greedy decoder's filters + update on caller-supplied logits (kernel parity test) */
int wca_test_decode_select(wca_engine* e, const float* logits_dev, int batch, int n_vocab, int32_t* tokens_dev, int T_max,
                           int cur_len, int n_initial, const uint8_t* suppress_mask_dev, const uint8_t* blank_mask_dev,
                           const wca_decode_opts* opts, float* sum_logprob_dev, int32_t* n_done_dev);
int wca_test_layernorm(wca_engine* e, const float* x_dev, const float* g_dev, const float* b_dev, void* out_f16_dev,
                       int rows, int d);
/* split-mode LayerNorm: out2 f16 [rows][2d], hi = f16(y) at column c, lo = f16(y - hi) at column d + c */
int wca_test_layernorm_split(wca_engine* e, const float* x_dev, const float* g_dev, const float* b_dev, void* out2_f16_dev,
                             int rows, int d);
/* encoder only: mel_dev [batch][n_mels][3000] f32 -> xa_out_dev [batch][1500][d] f32 (ln_post output) */
int wca_test_encoder(wca_engine* e, const float* mel_dev, int batch, float* xa_out_dev);

/* timing of the last wca_align_batch* call, milliseconds per stage measured with HIP events on the
 * engine stream: [0] log-mel, [1] encoder, [2] cross K/V projection, [3] decoder, [4] head stats,
 * [5] top-k + aggregate, [6] DTW, [7] total. Valid after a synchronize/fetch. */
int wca_last_stage_ms(wca_engine* e, float* ms8);
/* Per-kernel timing of the encoder layers in the last wca_align_batch* call (profiling enabled): HIP-event pairs around
 * every launch of one kernel site, on the stream the kernel runs on. n_launches = encoder layers; flops / bytes = the
 * ALGORITHMIC work of one launch at the last batch size (M = batch * 1500 rows; bytes: operands read once, outputs
 * written once, f32 residual read + written for the two read-modify-write GEMMs). */
enum {
  WCA_SITE_QKV = 0,   /* encoder self-attention q/k/v projection   gemm256p<0,false,1>  N = 3d, K = d   */
  WCA_SITE_ATTN = 1,  /* encoder self-attention (flash)            attn32_kernel<false>                 */
  WCA_SITE_OUT = 2,   /* attention out-projection + residual: gemm256p<2,false,1> alone (or the fused gemm256p<3,..> + mlp_ln)  */
  WCA_SITE_FC1 = 3,   /* MLP fc1 + GELU                            gemm256p<0,true,1>   N = 4d, K = d   */
  WCA_SITE_FC2 = 4,   /* MLP fc2 + residual: gemm256p<2,false,4> alone (or fused with the next attn_ln / ln_post)                */
  WCA_SITE_LN1 = 5,   /* attn_ln launches (slot li = the layer they feed; slot n_layer = ln_post)                                */
  WCA_SITE_LN2 = 6,   /* mlp_ln launches                                                                                         */
  WCA_N_SITES = 7
};
int wca_last_kernel_ms(wca_engine* e, int site, int* n_launches, float* total_ms, double* flops_per_launch,
                       double* bytes_per_launch);
/* on: the encoder's LayerNorms run in the epilogue of the GEMM that produces their input (attention out-projection ->
 * mlp_ln, fc2 -> the next layer's attn_ln / ln_post) where the shape allows it (>= 192 tiles, N % 256 == 0; otherwise, and
 * always in split mode, the separate launches); off (default): separate LayerNorm launches everywhere. Results agree up to
 * fp32 rounding of the row statistics (a last-bit difference of a few f16 outputs per thousand). Measured at the bench
 * configuration: -0.15 ms per encoder layer at kernel level, +0.9 % end to end; bench.py turns it on.
 * REQUIRES THE GPU TO ITSELF: the N / 256 workgroups of a 256-row panel wait for each other inside the launch, so all of them
 * must be resident together. Kernels of the same engine never prevent that for long (they are short and finite), but a second
 * process or engine running the same kind of launch on the device can hold the CUs its siblings need: the spin is bounded
 * (seconds) and the batch then fails with WCA_ERR_HIP instead of hanging (observed with two bench ranks sharing one GPU). */
int wca_set_fuse_ln(wca_engine* e, int on);
/* Arithmetic of the model forward (reference: timing.py:58, `model(mel.unsqueeze(0), tokens.unsqueeze(0))` -- an fp32
 * forward of a checkpoint whose weights are f16 at rest):
 *   WCA_PRECISION_F16 (opt-in since round 5: wca_engine_create_ex / wca_set_precision): GEMM / attention operands are rounded to f16 once (11 significant bits), accumulation,
 *     residual stream, LayerNorm, softmax and everything downstream fp32. Fastest; attention maps agree with the fp32
 *     reference to ~3e-3, which can move an ill-conditioned DTW path or swap two near-tied heads of the top-k selection.
 *   WCA_PRECISION_SPLIT: reference precision on the f16 matrix pipe. Every activation operand x travels as the pair
 *     hi = f16(x), lo = f16(x - hi) (x = hi + lo to 2^-22 |x|) in one row [hi | lo]; weights are exact in f16, so
 *     A W^T = [A_hi | A_lo] [W | W]^T is a K-doubled call of the same MFMA GEMM kernels (f16 x f16 products are exact in
 *     the fp32 accumulator); attention runs three passes per product (hi.hi + hi.lo + lo.hi), GELU uses erff, the log-mel
 *     DFT accumulates in f64. What is left is fp32 summation-order noise, like between two fp32 BLAS libraries. Costs
 *     ~2.3x the MFMA work, twice the operand memory and a second copy of the weights ([N][2K]).
 * The greedy ASR pre-pass (wca_greedy_decode) computes in f16 in both modes, like whisper.decode's fp16 default.
 * Switching re-creates the activation arena: no batch may be in flight, encoded-but-unconsumed states are dropped. */
/*   WCA_PRECISION_REFERENCE = WCA_PRECISION_SPLIT: the CONTRACT mode -- every site split. A new engine (wca_engine_create), bench.py's
 *     `value` and the CLI run in this mode. The per-site ablation on the 301-utterance parity leg (profiles/r04_precision_ablation.txt, tools/precision_ablation.py)
 *     shows that nothing from the encoder blocks on can be left on single f16 operands: every smaller site set misses at least one
 *     utterance whose 10th / 11th oracle head scores are within 5e-5 of each other. Leaving only the log-mel and the conv stem on
 *     single operands changes no boundary on that leg and costs 1 % less, but moves the selection scores by 1e-4 relative
 *     (all sites split: 4e-6), and on a second leg of 700 utterances (profiles/r04_parity_leg_700utt.txt) it misses two utterances
 *     whose oracle scores are tied to 1e-5 while all sites give 14 234 / 14 234 boundaries identical: it is not the contract. */
enum { WCA_PRECISION_F16 = 0, WCA_PRECISION_SPLIT = 1, WCA_PRECISION_REFERENCE = 1, WCA_PRECISION_MIXED = 2 };
int wca_set_precision(wca_engine* e, int mode);   /* F16 or SPLIT (= REFERENCE) */
int wca_get_precision(wca_engine* e);  /* F16: no site is split; SPLIT: every site; MIXED: some (wca_get_precision_sites) */
/* Per-site precision control: WHICH stages of the forward of timing.py:58 carry their operands as (hi, lo) pairs. A set bit
 * puts that stage on reference-precision arithmetic (above); a clear bit leaves it on single f16 operands. Seams need no
 * conversion pass: a producer stores the pair when its consumer is split (every GEMM / LayerNorm / log-mel epilogue can), a
 * single-precision consumer of a pair buffer reads the hi halves (hi IS f16(x)), and a split GEMM behind a single-precision
 * attention reads the single f16 rows it got.
 *   LOGMEL     log-mel DFT + filterbank sums in f64 (dataset.py:46-48 -> whisper.log_mel_spectrogram)
 *   CONV       conv stem: conv1 + GELU, conv2 + GELU + positional embedding (AudioEncoder.forward)
 *   ENC_GEMM   encoder blocks >= enc_first_layer: attn_ln / mlp_ln outputs as pairs, QKV / out-projection / fc1 / fc2 K-doubled
 *   ENC_ATTN   encoder blocks >= enc_first_layer: three-pass self-attention on q / k / v pairs
 *   CROSS_KV   ln_post output as pairs + the fused cross-attention key / value projection of every decoder layer K-doubled
 *   DEC        teacher-forced decoder: its LayerNorms, QKV / out / cross-query / cross-out / MLP GEMMs and causal self-attention
 *   CAPTURE    the hooked cross-attention (timing.py:50-55): q and the cross K / V rows as pairs, three-pass q.k^T (the captured
 *              logits) and P.V
 * wca_set_precision(SPLIT) == wca_set_precision_sites(WCA_PSITE_ALL, 0); F16 == mask 0. enc_first_layer in [0, n_audio_layer].
 * Same state rules as wca_set_precision (no batch in flight). The arena is widened (and the K-doubled weight copies exist)
 * whenever any bit is set. */
enum {
  WCA_PSITE_LOGMEL = 1,
  WCA_PSITE_CONV = 2,
  WCA_PSITE_ENC_GEMM = 4,
  WCA_PSITE_ENC_ATTN = 8,
  WCA_PSITE_CROSS_KV = 16,
  WCA_PSITE_DEC = 32,
  WCA_PSITE_CAPTURE = 64,
  WCA_PSITE_ALL = 127
};
int wca_set_precision_sites(wca_engine* e, unsigned mask, int enc_first_layer);
int wca_get_precision_sites(wca_engine* e, unsigned* mask_out, int* enc_first_layer_out);
/* on (default): phase 2 (decoder, post-processing, DTW) of a batch runs on the engine's second stream beside the next
 * batch's phase 1; off: everything on one stream, so that rocprofv3 per-kernel durations are not inflated by sharing
 * the CUs (profiling aid; throughput drops by the overlap's worth). No batch may be in flight when it is changed. */
int wca_set_overlap(wca_engine* e, int on);
/* EXPERIMENT (measured in profiles/r04_cu_partition.txt, off by default): phase2_cus > 0 (a multiple of 8) gives phase 2
 * of the alignment and the greedy decode loop CU-masked streams that own that many compute units (hipExtStreamCreateWithCUMask) and phase 1
 * (log-mel, encoder, cross-K/V) the rest, its persistent GEMM grids sized to match -- so that the HBM-bound decode kernels run BESIDE
 * the MFMA-bound encoder instead of queueing behind its persistent workgroups. 0 lifts the partition (phase 1 returns to the stream the caller
 * bound last). While it is active wca_engine_set_stream records the stream and returns WCA_STATUS_PARTITIONED (the caller's stream has no
 * mask): inputs must be complete before an entry point is called. Not together with wca_set_fuse_ln (WCA_ERR_STATE). A failure while creating
 * the masked streams leaves the previous partition (or none) in place. */
int wca_set_cu_partition(wca_engine* e, int phase2_cus);
/* Decoder GEMMs on few rows (a greedy-decode step of wca_greedy_decode: M = batch; batch-1 teacher-forced forwards):
 * fused != 0 (default): for up to 128 rows one few-row kernel per GEMM with the LayerNorm in its prologue, the KV-cache append in its
 * epilogue and deterministic split-K for K = 4 n_state; 0: separate LayerNorm / GEMM / append launches (the round-1 path,
 * kept for A/B tests). streams: 1 (default) or 2 = the batch as two half-batches enqueued layer by layer in turn on two streams
 * (measured on MI355X: the two hardware queues' kernels run back to back rather than concurrently, 3.86 vs 3.90 ms per step).
 * Token choices are independent of both settings up to fp32 summation order in the split-K GEMM. */
int wca_set_decode_mode(wca_engine* e, int fused, int streams);
/* enable/disable per-stage event recording (default off: no extra events on the stream) */
int wca_set_profiling(wca_engine* e, int on);

#ifdef __cplusplus
}
#endif
#endif /* WCA_H_ */
