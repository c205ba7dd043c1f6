// libwca.so engine: weight residency, activation arena, forward orchestration and the C ABI of
// include/wca.h.  One engine = one MI355X = one host thread; two HIP streams (phase 1: log-mel, encoder, cross-K/V;
// phase 2: decoder / greedy decode loop, post-processing, DTW, D2H) with two cross-K/V slots between them.
//
// Data layout in HBM (per engine, B = max_batch, d = n_state, L = decoder layers):
//   weights   f16 [N][K] row-major (torch Linear layout), q/k/v fused to [3d][d]; the cross-attention
//             key/value projections of ALL decoder layers fused to one [L*2*d][d] matrix so that the
//             encoder output is projected by a single large MFMA GEMM; conv kernels re-ordered to
//             [d][tap*C + c] so the conv stem is a GEMM over overlapping time-major windows.
//   residual  f32 [B*1500][d] (encoder), f32 [B*n][d] (decoder); GEMM operands are f16.
//   capture   f32 [B][L*H][n_max][Fpad]  pre-softmax cross-attention logits (timing.py:50-55)
//   weights_ws f32 [B][L*H][n_max][Fmax] filtered+softmaxed maps (timing.py:63-66; step-by-step API only)
//   decode    f16 self-attention K/V cache [L][2][B][T_max][d], int32 token rows, fp32 logits [B][n_vocab]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/wca.h"
#include "kernels.h"

using namespace wca;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) return fail(WCA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

constexpr int N_FRAMES = 3000, N_CTX = 1500, MAX_TOK = 448, N_BIN = 201, N_FFT = 400;
constexpr int META_SLOTS = 16;
constexpr int DEC_ROWS_MAX = 128;  // decoder GEMMs on at most this many rows take the few-row kernel (gemm_rows.hip)

#define WCA_TRY(expr)          \
  do {                         \
    const int _rc = (expr);    \
    if (_rc != WCA_OK) return _rc; \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct GrowBuf {
  void* p = nullptr;
  size_t bytes = 0;
  hipError_t ensure(size_t need) {
    if (need <= bytes) return hipSuccess;
    if (p) {
      hipError_t e = hipFree(p);
      if (e != hipSuccess) return e;
      p = nullptr;
      bytes = 0;
    }
    need = align_up(need, 1 << 20);
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) return e;
    bytes = need;
    // debugging aid: WCA_POISON_ALLOC=1 fills every grow-only buffer with 0xFF bytes (NaN as f32 / f16, -1 as an index) when it is
    // allocated, so that a read of a never-written element shows as a wrong result on every run instead of depending on what the
    // recycled memory held
    static const bool poison = std::getenv("WCA_POISON_ALLOC") != nullptr;
    if (poison) {
      e = hipMemset(p, 0xFF, need);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

struct LayerW {
  float *ln1_g, *ln1_b;
  half_t* qkv_w;
  float* qkv_b;
  half_t* out_w;
  float* out_b;
  float *ln2_g, *ln2_b;  // mlp_ln
  half_t* fc1_w;
  float* fc1_b;
  half_t* fc2_w;
  float* fc2_b;
  // decoder only
  float *lnc_g, *lnc_b;
  half_t* cq_w;
  float* cq_b;
  half_t* co_w;
  float* co_b;
};

}  // namespace

struct wca_engine {
  wca_model_dims dims;
  int device = 0;
  int max_batch = 1;
  hipStream_t stream = nullptr;      // phase 1 (log-mel, encoder, cross-K/V) and every non-batched entry point
  hipStream_t own_stream = nullptr;
  hipStream_t stream2 = nullptr;     // phase 2 of wca_align_batch (decoder, post-processing, DTW, D2H): overlaps the next batch's phase 1
  hipStream_t stream3 = nullptr;     // second half-batch of the greedy decode loop (wca_greedy_decode): its latency-bound small
                                     // kernels run under the other half's HBM-bound cross-attention
  hipEvent_t ev_kv[2] = {};          // cross-K/V of batch slot ready (recorded on `stream`)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;  // stream2 -> stream3 fork / join of the decode loop
  bool finalized = false;
  bool have_filters = false;
  bool profiling = false;
  std::set<std::string> loaded;
  std::map<std::string, size_t> inexact;  // tensors stored as f16 whose fp32 source values were NOT f16-representable: name -> count of rounded elements
  bool allow_rounded = false;             // wca_set_allow_rounded_weights: run pair sites on the ROUNDED weights (faster; not the fp32 model's arithmetic)
  char* wslab_lo = nullptr;               // W_lo slab, same layout as wslab (allocated when the first inexact tensor arrives): lo = f16(w - f16(w)) of every
                                          // weight matrix element, zero where the f16 value is exact. A pair site multiplies the extra term A_hi W_lo^T
  std::set<const void*> wlo_bases;        // weight matrices (base pointer the GEMM call sites use) that hold at least one non-zero lo element
  GrowBuf wlo_tmp[2];                     // f32 [M][N] scratch of the extra term for the non-accumulating output modes: [0] launches on `stream` (phase 1),
                                          // [1] on any other stream (phase 2 runs beside the next batch's phase 1)

  // ---- weights
  char* wslab = nullptr;
  size_t wslab_bytes = 0, wslab_used = 0;
  int k1pad = 0;  // padded K of the conv1 GEMM
  half_t *conv1_w = nullptr, *conv2_w = nullptr;
  float *conv1_b = nullptr, *conv2_b = nullptr, *enc_pos = nullptr, *lnpost_g = nullptr, *lnpost_b = nullptr;
  std::vector<LayerW> enc, dec;
  half_t* tok_emb = nullptr;
  float *dec_pos = nullptr, *lnf_g = nullptr, *lnf_b = nullptr;
  half_t* kv_w = nullptr;
  float* kv_b = nullptr;
  float *mel_filters = nullptr, *window = nullptr, *twiddle = nullptr;
  int *filt_lo = nullptr, *filt_hi = nullptr;

  // ---- fixed activation arena (sized for max_batch)
  char* aslab = nullptr;
  float* mel_scratch = nullptr;
  unsigned* gmax = nullptr;
  float* mel_f32 = nullptr;
  half_t* mel_tm = nullptr;
  half_t* h1pad = nullptr;
  float* x = nullptr;      // [B*1500][d]
  half_t* xn = nullptr;    // [B*1500][d]
  half_t* qkv = nullptr;   // [B*1500][3d]
  half_t* att = nullptr;   // [B*1500][d]
  half_t* hid = nullptr;   // [B*1500][4d]
  half_t* kv = nullptr;    // [B*1500][L*2*d]
  half_t* kv_alt = nullptr; // second cross-K/V buffer (batches alternate, see wca_align_batch_enqueue)
  float* xd = nullptr;     // [B*448][d]
  half_t* xdn = nullptr;
  half_t* qkv_d = nullptr;
  half_t* att_d = nullptr;
  half_t* q_d = nullptr;
  half_t* hid_d = nullptr;
  int* meta_dev = nullptr;   // META_SLOTS x 4 x max_batch ints: n_samples, n_tok, n_frames, dtwN
  int* meta_host = nullptr;  // pinned mirror
  int meta_slot = 0;
  unsigned long long* ln_stats = nullptr;  // out_mode 3 GEMMs: per-tile row statistics [n_state/256][B*1500 padded to 256]
  unsigned* ln_cnt = nullptr;              // ... and per-panel arrival counters (zeroed by launch_gemm)
  float* sk_part[2] = {nullptr, nullptr};  // split-K workspaces of the few-row GEMM (one per decode stream) and their
  unsigned* sk_cnt[2] = {nullptr, nullptr};  // arrival counters (zero at creation, self-cleaning)
  size_t sk_floats = 0, sk_tiles = 0;
  float* sk_big[3] = {nullptr, nullptr, nullptr};   // split-K partials of the 128 x 128 tile GEMM (small batches: fc2): [0] encoder stream,
                                                     // [1 + ws] decoder / decode stream ws (the two half-batches of the decode loop may run it concurrently)
  size_t sk_big_bytes = 0;
  int n_cu = 0;
  int* err_dev = nullptr;    // device flags raised by kernels. Word 0: phase 2 / synchronous entry points (bit 0 token id outside the
                             // vocabulary, bit 1 LayerNorm hand-off timeout); words 1, 2: phase 1 of the batch in cross-K/V slot 0, 1 (bit 1
                             // only), cleared on `stream` before that batch's encoder and read with the batch's results, so that neither the
                             // phase-2 clear of this batch nor the next batch's encoder (concurrent on `stream`) can wipe or alias it
  int* ln_err = nullptr;     // where the encoder's out_mode-3 GEMMs raise their time-out bit (err_dev, or err_dev + 1 + slot in run_phase1)
  int* err_host = nullptr;   // pinned: read back by the synchronous entry points

  // ---- run-time sized buffers
  GrowBuf cap, wws, colnorm, scores, sel, selsc, matrix, trace, path, pathlen, jump, tmp0, tmp1;
  GrowBuf probe_jump;          // the last wca_probe_heads' jump frames [LH][N], kept on the device for wca_probe_strict_tp
  int probe_LH = 0, probe_N = 0;
  // greedy ASR pre-pass (wca_greedy_decode): self-attention K/V cache [L][2][B][T_max][d], token rows, masks, logits
  GrowBuf dec_cache, dec_tokens, dec_masks, dec_logits, dec_state;
  int* dec_done_host = nullptr;  // pinned: completion counter read back while the loop runs
  // Encoded micro-batches (log-mel + encoder + cross-K/V done, recorded on `stream`) that no alignment has consumed
  // yet: wca_encode_batch / wca_greedy_decode push, wca_align_batch_enqueue(pcm_dev = NULL) pops the oldest. A K/V
  // slot stays busy from its encode until the alignment that consumed it has been fetched.
  struct EncState { int slot; int batch; bool decoded; };
  std::deque<EncState> enc_q;
  bool slot_busy[2] = {false, false};
  int res_kvslot[2] = {-1, -1};
  // results ring: up to 2 wca_align_batch_enqueue calls may be in flight before their _fetch
  int* res_host[2] = {nullptr, nullptr};  // pinned results staging
  size_t res_host_ints[2] = {0, 0};
  hipEvent_t res_ev[2] = {};
  int res_topk[2] = {0, 0}, res_ntok[2] = {0, 0}, res_batch[2] = {0, 0};
  unsigned long enq_count = 0, fetch_count = 0;
  int last_batch = 0;

  hipEvent_t ev[9] = {};
  // start/stop pairs around each kernel of every encoder layer (profiling only): site = WCA_SITE_* of include/wca.h
  hipEvent_t kev[WCA_N_SITES][33][2] = {};   // slot 32: ln_post of a 32-layer encoder (site LN1, slot n_layer)
  bool kev_set[WCA_N_SITES][33] = {};   // which (site, layer) pairs the last encoder run recorded
  // ---- reference-precision ("split") mode, wca_set_precision: every f16 GEMM / attention operand x travels as the pair
  // hi = f16(x), lo = f16(x - hi) in ONE row [hi(K) | lo(K)], and every weight matrix as [W | W] ([N][2K], built once on the
  // device from the f16 weights, which are exact): A.W^T = [A_hi | A_lo].[W | W]^T is then a K-doubled call of the SAME GEMM
  // kernels with f16 x f16 products exact in the fp32 accumulator. Activation operand buffers are twice as wide.
  bool split = false;        // some site is split (sites != 0): the arena is wide and the K-doubled weight copies exist
  unsigned sites = 0;        // WCA_PSITE_* bits: which stages run on (hi, lo) operand pairs (wca_set_precision_sites)
  int enc_from = 0;          // encoder blocks >= enc_from take the ENC_GEMM / ENC_ATTN bits
  char* wslab2 = nullptr;    // the K-doubled weight copies (allocated while split is on)
  bool sw_dirty = true;      // a weight was (re)loaded since the copies were built
  struct SplitW {
    half_t *conv1_w = nullptr, *conv2_w = nullptr, *kv_w = nullptr, *tok_emb = nullptr;
    half_t *conv1_wlo = nullptr, *conv2_wlo = nullptr;   // [W_lo | 0] per tap group: the conv stem's extra term against [hi(C) | lo(C)] frames (inexact conv weights)
    int k1pad = 0;           // padded K of the split conv1 GEMM: windows of 3 frames x [hi(C) | lo(C)]
    std::vector<LayerW> enc, dec;  // only the half_t* members are used
  } sw;
  // ---- collation over RCCL (wca_comm_init / wca_allgather_results): this engine's rank in a communicator of one rank per GPU
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 0;
  GrowBuf coll_send, coll_recv;
  bool fuse_ln = false;      // LayerNorm in the epilogue of the residual GEMMs where the shape allows (wca_set_fuse_ln; never in split
                             // mode). OFF by default: the workgroups of a row panel wait for each other inside the launch, which needs
                             // the GPU to itself -- with a second process (or engine) on the device two such launches can hold each
                             // other's CUs and run into the bounded spin's time-out (measured: two bench ranks on one GPU)
  bool overlap = true;       // phase 2 on its own stream (false: everything on `stream`, for clean per-kernel profiles)
  int part_cus = 0;          // wca_set_cu_partition: > 0 = phase 2 / the decode loop own that many CUs (CU-masked streams), phase 1 the rest
  hipStream_t part_s1 = nullptr, part_s2 = nullptr, part_s3 = nullptr;  // the masked streams: they REPLACE stream / stream2 / stream3 while active
  hipStream_t saved_s2 = nullptr, saved_s3 = nullptr;                   // ... and the engine's own ones come back when the partition is lifted
  hipStream_t user_stream = nullptr;   // the stream the caller bound last (wca_engine_set_stream), also while a partition is active
  bool user_stream_set = false;
  bool dec_fused = true;     // few-row GEMM with LayerNorm prologue / KV append / split-K for M <= DEC_ROWS_MAX = 128 rows (wca_set_decode_mode)
  int dec_streams = 1;       // 2: the greedy decode loop as two half-batches on two streams (measured: the two queues' kernels run
                             // back to back, not concurrently -- 3.86 vs 3.90 ms per step -- so one stream is the default)
  bool ev_valid = false;
  float stage_ms[8] = {};
};

namespace {

template <typename T>
T* carve(char*& cur, size_t count, size_t align = 256) {
  uintptr_t p = reinterpret_cast<uintptr_t>(cur);
  p = (p + align - 1) / align * align;
  T* r = reinterpret_cast<T*>(p);
  cur = reinterpret_cast<char*>(p + count * sizeof(T));
  return r;
}

// ---- weights that are not exact in f16 (fp32 checkpoints): the W_lo slab mirrors wslab byte for byte
inline bool use_wlo(const wca_engine* e) { return e->wslab_lo != nullptr && !e->wlo_bases.empty() && !e->allow_rounded; }
inline const half_t* wlo_of(const wca_engine* e, const half_t* w) {
  return reinterpret_cast<const half_t*>(e->wslab_lo + (reinterpret_cast<const char*>(w) - e->wslab));
}

// ---- per-site precision (wca_set_precision_sites): which stages compute on (hi, lo) operand pairs
inline bool site_on(const wca_engine* e, unsigned bit) { return (e->sites & bit) != 0; }
inline bool enc_gemm_split(const wca_engine* e, int li) { return (e->sites & WCA_PSITE_ENC_GEMM) && li >= e->enc_from && li < e->dims.n_audio_layer; }
inline bool enc_attn_split(const wca_engine* e, int li) { return (e->sites & WCA_PSITE_ENC_ATTN) && li >= e->enc_from && li < e->dims.n_audio_layer; }

// Operands of one GEMM at a seam. a_pair: the A buffer holds [hi(K) | lo(K)] rows (row stride 2 K); want: the GEMM's site is
// split. Both: the K-doubled product [A_hi | A_lo] [W | W]^T. A single-precision site behind a pair producer reads the hi halves
// (hi IS f16(x)); a split site behind a single-precision producer multiplies the f16 rows it got (there is no lo to add).
// Where launch_gemm takes the persistent 256 x 256 kernel (M, N given), the pair product runs in its SPLITW form: plain W, each W
// K-tile staged once (a_lo = K); elsewhere as the K-doubled call on the [W | W] copy.
struct GemmOpnd {
  const half_t* W;
  int lda, K, ldw;
  long a_lo;
  const half_t* Wp;   // the plain [N][K] matrix when the product is a PAIR product (null otherwise): where the W_lo term of an inexact matrix comes from
  int Kp;
};
inline GemmOpnd pick_operands(bool a_pair, bool want, const half_t* W1, const half_t* W2, int K, int M = 0, int N = 0, int out_mode = 0, const wca_engine* wlo_e = nullptr) {
  const bool use = a_pair && want;
  // (a matrix with a W_lo remainder takes the K-doubled call for its non-accumulating products: their extra term enters through the generic epilogue's addend)
  const bool no_splitw = wlo_e != nullptr && out_mode != 2 && use_wlo(wlo_e) && wlo_e->wlo_bases.count(W1) != 0;
  if (use && !no_splitw && M > 0 && gemm_splitw_supported(M, N, K, 2 * K, out_mode)) return GemmOpnd{W1, 2 * K, K, K, (long)K, W1, K};
  return GemmOpnd{use ? W2 : W1, a_pair ? 2 * K : K, use ? 2 * K : K, use ? 2 * K : K, 0, use ? W1 : nullptr, K};
}

// ---- weight slab layout (two passes: size, then carve)
size_t layout_weights(wca_engine* e, char* base) {
  const wca_model_dims& D = e->dims;
  const int d = D.n_audio_state, dt = D.n_text_state;
  char* cur = base;
  e->k1pad = (int)align_up((size_t)3 * D.n_mels, 64);
  e->conv1_w = carve<half_t>(cur, (size_t)d * e->k1pad);
  e->conv1_b = carve<float>(cur, d);
  e->conv2_w = carve<half_t>(cur, (size_t)d * 3 * d);
  e->conv2_b = carve<float>(cur, d);
  e->enc_pos = carve<float>(cur, (size_t)N_CTX * d);
  e->lnpost_g = carve<float>(cur, d);
  e->lnpost_b = carve<float>(cur, d);
  e->enc.resize(D.n_audio_layer);
  for (auto& l : e->enc) {
    l.ln1_g = carve<float>(cur, d);
    l.ln1_b = carve<float>(cur, d);
    l.qkv_w = carve<half_t>(cur, (size_t)3 * d * d);
    l.qkv_b = carve<float>(cur, 3 * d);
    l.out_w = carve<half_t>(cur, (size_t)d * d);
    l.out_b = carve<float>(cur, d);
    l.ln2_g = carve<float>(cur, d);
    l.ln2_b = carve<float>(cur, d);
    l.fc1_w = carve<half_t>(cur, (size_t)4 * d * d);
    l.fc1_b = carve<float>(cur, 4 * d);
    l.fc2_w = carve<half_t>(cur, (size_t)4 * d * d);
    l.fc2_b = carve<float>(cur, d);
    l.lnc_g = l.lnc_b = nullptr;
    l.cq_w = l.co_w = nullptr;
    l.cq_b = l.co_b = nullptr;
  }
  e->tok_emb = carve<half_t>(cur, (size_t)D.n_vocab * dt);
  e->dec_pos = carve<float>(cur, (size_t)D.n_text_ctx * dt);
  e->lnf_g = carve<float>(cur, dt);
  e->lnf_b = carve<float>(cur, dt);
  e->kv_w = carve<half_t>(cur, (size_t)D.n_text_layer * 2 * dt * d);
  e->kv_b = carve<float>(cur, (size_t)D.n_text_layer * 2 * dt);
  e->dec.resize(D.n_text_layer);
  for (auto& l : e->dec) {
    l.ln1_g = carve<float>(cur, dt);
    l.ln1_b = carve<float>(cur, dt);
    l.qkv_w = carve<half_t>(cur, (size_t)3 * dt * dt);
    l.qkv_b = carve<float>(cur, 3 * dt);
    l.out_w = carve<half_t>(cur, (size_t)dt * dt);
    l.out_b = carve<float>(cur, dt);
    l.lnc_g = carve<float>(cur, dt);
    l.lnc_b = carve<float>(cur, dt);
    l.cq_w = carve<half_t>(cur, (size_t)dt * dt);
    l.cq_b = carve<float>(cur, dt);
    l.co_w = carve<half_t>(cur, (size_t)dt * dt);
    l.co_b = carve<float>(cur, dt);
    l.ln2_g = carve<float>(cur, dt);
    l.ln2_b = carve<float>(cur, dt);
    l.fc1_w = carve<half_t>(cur, (size_t)4 * dt * dt);
    l.fc1_b = carve<float>(cur, 4 * dt);
    l.fc2_w = carve<half_t>(cur, (size_t)4 * dt * dt);
    l.fc2_b = carve<float>(cur, dt);
  }
  e->mel_filters = carve<float>(cur, (size_t)D.n_mels * N_BIN);
  e->window = carve<float>(cur, N_FFT);
  e->twiddle = carve<float>(cur, 2 * N_FFT);
  e->filt_lo = carve<int>(cur, D.n_mels);
  e->filt_hi = carve<int>(cur, D.n_mels);
  return (size_t)(cur - base) + 4096;
}

size_t layout_arena(wca_engine* e, char* base) {
  const wca_model_dims& D = e->dims;
  const size_t B = e->max_batch, d = D.n_audio_state, dt = D.n_text_state, L = D.n_text_layer;
  const size_t om = e->split ? 2 : 1;  // f16 operand buffers hold [hi | lo] rows in split mode
  char* cur = base;
  e->mel_scratch = carve<float>(cur, B * D.n_mels * N_FRAMES);
  e->gmax = carve<unsigned>(cur, B);
  e->mel_f32 = carve<float>(cur, B * D.n_mels * N_FRAMES);
  e->mel_tm = carve<half_t>(cur, om * B * (N_FRAMES + 2) * D.n_mels + 4096);
  e->h1pad = carve<half_t>(cur, om * B * (N_FRAMES + 2) * d + 4096);
  e->x = carve<float>(cur, B * N_CTX * d);
  e->xn = carve<half_t>(cur, om * B * N_CTX * d);
  e->qkv = carve<half_t>(cur, om * B * N_CTX * 3 * d);
  e->att = carve<half_t>(cur, om * B * N_CTX * d);
  e->hid = carve<half_t>(cur, om * B * N_CTX * 4 * d);
  e->kv = carve<half_t>(cur, om * B * N_CTX * L * 2 * dt);
  e->kv_alt = carve<half_t>(cur, om * B * N_CTX * L * 2 * dt);
  e->xd = carve<float>(cur, B * MAX_TOK * dt);
  e->xdn = carve<half_t>(cur, om * B * MAX_TOK * dt);
  e->qkv_d = carve<half_t>(cur, om * B * MAX_TOK * 3 * dt);
  e->att_d = carve<half_t>(cur, om * B * MAX_TOK * dt);
  e->q_d = carve<half_t>(cur, om * B * MAX_TOK * dt);
  e->hid_d = carve<half_t>(cur, om * B * MAX_TOK * 4 * dt);
  e->meta_dev = carve<int>(cur, (size_t)META_SLOTS * 4 * B);
  e->err_dev = carve<int>(cur, 64);
  {
    const size_t mpad = align_up(B * N_CTX, 256);
    e->ln_stats = carve<unsigned long long>(cur, (d / 256 + 1) * mpad);
    e->ln_cnt = carve<unsigned>(cur, mpad / 256 + 16, 256);
  }
  {
    // few-row GEMM, split-K (K > 1024: fc2; every K of the 1280-wide model): up to ROWS_MAX rows, N = n_text_state columns
    const int s_max = std::max(std::max(gemm_rows_pick_splitk((int)dt), gemm_rows_pick_splitk((int)(4 * dt))), 1);
    e->sk_tiles = (size_t)(DEC_ROWS_MAX / 64) * ((dt + 15) / 16);
    e->sk_floats = e->sk_tiles * s_max * 64 * 16;
    for (int i = 0; i < 2; ++i) {
      e->sk_part[i] = carve<float>(cur, e->sk_floats);
      e->sk_cnt[i] = carve<unsigned>(cur, e->sk_tiles + 16, 256);
    }
    // 128 x 128 tile GEMM with at most n_cu / 2 = 128 tiles, 4 K slices of f32 partials
    e->sk_big_bytes = (size_t)4 * 128 * 128 * 128 * sizeof(float);
    for (int i = 0; i < 3; ++i) e->sk_big[i] = carve<float>(cur, e->sk_big_bytes / sizeof(float));
  }
  return (size_t)(cur - base) + 4096;
}

// ---- split mode: the K-doubled weight copies (two passes like layout_weights: size, then carve)
size_t layout_split_weights(wca_engine* e, char* base) {
  const wca_model_dims& D = e->dims;
  const size_t d = D.n_audio_state, dt = D.n_text_state;
  char* cur = base;
  e->sw.k1pad = (int)align_up((size_t)6 * D.n_mels, 64);
  e->sw.conv1_w = carve<half_t>(cur, d * e->sw.k1pad);
  e->sw.conv2_w = carve<half_t>(cur, d * 6 * d);
  e->sw.conv1_wlo = carve<half_t>(cur, d * e->sw.k1pad);
  e->sw.conv2_wlo = carve<half_t>(cur, d * 6 * d);
  e->sw.enc.assign(D.n_audio_layer, LayerW{});
  for (auto& l : e->sw.enc) {
    l.qkv_w = carve<half_t>(cur, 3 * d * 2 * d);
    l.out_w = carve<half_t>(cur, d * 2 * d);
    l.fc1_w = carve<half_t>(cur, 4 * d * 2 * d);
    l.fc2_w = carve<half_t>(cur, d * 8 * d);
  }
  e->sw.tok_emb = carve<half_t>(cur, (size_t)D.n_vocab * 2 * dt);
  e->sw.kv_w = carve<half_t>(cur, (size_t)D.n_text_layer * 2 * dt * 2 * d);
  e->sw.dec.assign(D.n_text_layer, LayerW{});
  for (auto& l : e->sw.dec) {
    l.qkv_w = carve<half_t>(cur, 3 * dt * 2 * dt);
    l.out_w = carve<half_t>(cur, dt * 2 * dt);
    l.cq_w = carve<half_t>(cur, dt * 2 * dt);
    l.co_w = carve<half_t>(cur, dt * 2 * dt);
    l.fc1_w = carve<half_t>(cur, 4 * dt * 2 * dt);
    l.fc2_w = carve<half_t>(cur, dt * 8 * dt);
  }
  return (size_t)(cur - base) + 4096;
}

// dst[n][(j / grp) * 2 * grp + (j % grp) + {0, grp}] = src[n][j] for j < K: every group of `grp` source columns is written twice,
// side by side. grp = K: [W | W] (a Linear weight against [hi(K) | lo(K)] activation rows); grp = channels of a conv input: the
// taps of the time-major conv GEMM against frames stored as [hi(C) | lo(C)]. Columns of dst past 2 K stay zero.
// second_zero: the second copy is zero -- [W_lo | 0]: a remainder matrix against pair rows multiplies the hi halves only
__global__ void dup_cols_kernel(const half_t* __restrict__ src, int ld_src, half_t* __restrict__ dst, int ld_dst, long N, int K, int grp, int second_zero) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * K) return;
  const long n = i / K;
  const int j = (int)(i - n * K);
  const half_t v = src[n * ld_src + j];
  const int g = j / grp, c = j - g * grp;
  half_t* o = dst + n * ld_dst + (long)g * 2 * grp + c;
  o[0] = v;
  o[grp] = second_zero ? (half_t)0.f : v;
}

int dup_cols(hipStream_t s, const half_t* src, int ld_src, half_t* dst, int ld_dst, long N, int K, int grp, int second_zero = 0) {
  const long tot = N * K;
  hipLaunchKernelGGL(dup_cols_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, ld_src, dst, ld_dst, N, K, grp, second_zero);
  HIPCHK(hipGetLastError());
  return WCA_OK;
}

// (re)build the K-doubled copies from the resident f16 weights when a weight changed since the last build
int ensure_split_weights(wca_engine* e) {
  if (!e->split || !e->sw_dirty) return WCA_OK;
  const wca_model_dims& D = e->dims;
  const int d = D.n_audio_state, dt = D.n_text_state, C = D.n_mels;
  hipStream_t s = e->stream;
  WCA_TRY(dup_cols(s, e->conv1_w, e->k1pad, e->sw.conv1_w, e->sw.k1pad, d, 3 * C, C));
  WCA_TRY(dup_cols(s, e->conv2_w, 3 * d, e->sw.conv2_w, 6 * d, d, 3 * d, d));
  if (e->wslab_lo) {   // the conv stem's remainder matrices in the layout of its pair GEMM
    WCA_TRY(dup_cols(s, wlo_of(e, e->conv1_w), e->k1pad, e->sw.conv1_wlo, e->sw.k1pad, d, 3 * C, C, 1));
    WCA_TRY(dup_cols(s, wlo_of(e, e->conv2_w), 3 * d, e->sw.conv2_wlo, 6 * d, d, 3 * d, d, 1));
  }
  for (int li = 0; li < D.n_audio_layer; ++li) {
    const LayerW& l = e->enc[li];
    const LayerW& w = e->sw.enc[li];
    WCA_TRY(dup_cols(s, l.qkv_w, d, w.qkv_w, 2 * d, 3 * d, d, d));
    WCA_TRY(dup_cols(s, l.out_w, d, w.out_w, 2 * d, d, d, d));
    WCA_TRY(dup_cols(s, l.fc1_w, d, w.fc1_w, 2 * d, 4 * d, d, d));
    WCA_TRY(dup_cols(s, l.fc2_w, 4 * d, w.fc2_w, 8 * d, d, 4 * d, 4 * d));
  }
  WCA_TRY(dup_cols(s, e->tok_emb, dt, e->sw.tok_emb, 2 * dt, D.n_vocab, dt, dt));
  WCA_TRY(dup_cols(s, e->kv_w, d, e->sw.kv_w, 2 * d, (long)D.n_text_layer * 2 * dt, d, d));
  for (int li = 0; li < D.n_text_layer; ++li) {
    const LayerW& l = e->dec[li];
    const LayerW& w = e->sw.dec[li];
    WCA_TRY(dup_cols(s, l.qkv_w, dt, w.qkv_w, 2 * dt, 3 * dt, dt, dt));
    WCA_TRY(dup_cols(s, l.out_w, dt, w.out_w, 2 * dt, dt, dt, dt));
    WCA_TRY(dup_cols(s, l.cq_w, dt, w.cq_w, 2 * dt, dt, dt, dt));
    WCA_TRY(dup_cols(s, l.co_w, dt, w.co_w, 2 * dt, dt, dt, dt));
    WCA_TRY(dup_cols(s, l.fc1_w, dt, w.fc1_w, 2 * dt, 4 * dt, dt, dt));
    WCA_TRY(dup_cols(s, l.fc2_w, 4 * dt, w.fc2_w, 8 * dt, dt, 4 * dt, 4 * dt));
  }
  // the copies are read by kernels on stream2 / stream3 too: make them visible before anything else is enqueued
  HIPCHK(hipStreamSynchronize(s));
  e->sw_dirty = false;
  return WCA_OK;
}

// c_lo > 0 (split mode, f16 output): the value is stored as the pair hi at C, lo at C + c_lo (out_mode 4)
thread_local int g_gemm_cu_limit = 0;  // CUs owned by the stream the current phase launches on (0 = the whole device)

// wlo_e / w_plain / k_plain (pair products only, GemmOpnd::Wp / Kp): when the engine holds a W_lo slab and this matrix has non-zero lo elements (an fp32
// checkpoint that is not exact in f16), the product gets its third term A_hi W_lo^T -- the A_lo W_lo^T term is below 2^-22 of the result like every dropped
// lo.lo term: an accumulating launch for the read-modify-write mode, an f32 scratch added before the activation (GemmArgs.addend) for the others.
hipError_t gemm(hipStream_t s, const half_t* A, int lda, const half_t* W, int ldw, const float* bias, void* C, int ldc, int M,
                int N, int K, int gelu, int out_mode, int site = 0, float* sk_ws = nullptr, size_t sk_bytes = 0, long c_lo = 0, long a_lo = 0,
                wca_engine* wlo_e = nullptr, const half_t* w_plain = nullptr, int k_plain = 0) {
  GemmArgs g{};
  g.a_lo = a_lo;
  g.cu_limit = g_gemm_cu_limit;
  g.sk_part = sk_ws;
  g.sk_bytes = sk_bytes;
  g.c_lo = c_lo;
  if (c_lo > 0 && out_mode == 0) out_mode = 4;
  g.A = A;
  g.lda = lda;
  g.W = W;
  g.ldw = ldw;
  g.bias = bias;
  g.C = C;
  g.ldc = ldc;
  g.M = M;
  g.N = N;
  g.K = K;
  g.gelu = gelu;
  g.out_mode = out_mode;
  g.site = site;
  if (wlo_e != nullptr && w_plain != nullptr && use_wlo(wlo_e) && wlo_e->wlo_bases.count(w_plain)) {
    GemmArgs x{};   // A_hi (the hi halves of the pair rows: same row stride, k_plain columns) x W_lo^T (plain [N][k_plain])
    x.cu_limit = g_gemm_cu_limit;
    x.A = A;
    x.lda = lda;
    x.W = wlo_of(wlo_e, w_plain);
    x.ldw = k_plain;
    x.M = M;
    x.N = N;
    x.K = k_plain;
    x.site = site;
    if (out_mode == 2) {
      x.C = C;
      x.ldc = ldc;
      x.out_mode = 2;
    } else {
      GrowBuf& tmp = wlo_e->wlo_tmp[s == wlo_e->stream ? 0 : 1];
      if (hipError_t he = tmp.ensure((size_t)M * N * sizeof(float)); he != hipSuccess) return he;
      x.C = tmp.p;
      x.ldc = N;
      x.out_mode = 1;
      g.addend = (const float*)tmp.p;
      g.ld_addend = N;
    }
    if (hipError_t he = launch_gemm(x, s); he != hipSuccess) return he;
  }
  return launch_gemm(g, s);
}

// One GEMM of the decoder on few rows (a greedy-decode step: M = batch; batch-1 teacher-forced forwards: M = n tokens).
// xln != nullptr: the A operand is LayerNorm(xln rows; ln_g, ln_b). kv_k != nullptr (QKV projection of a decode step, N = 3 d):
// the k / v columns go to the self-attention cache at position kv_t. M <= DEC_ROWS_MAX and a shape the few-row kernel takes:
// one launch (gemm_rows.hip); otherwise the separate LayerNorm / GEMM / kv_append launches.
int dec_gemm(wca_engine* e, hipStream_t s, int ws, const half_t* A, int lda, const float* xln, const float* ln_g, const float* ln_b,
             half_t* xn_scratch, const half_t* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K, int gelu, int out_mode,
             int site, half_t* kv_k = nullptr, half_t* kv_v = nullptr, int T_max = 0, int kv_t = 0) {
  const bool ln = xln != nullptr;
  const int sk = gemm_rows_pick_splitk(K);
  const bool fits = sk <= 1 || (kv_k == nullptr && (size_t)((M + 63) / 64) * ((N + 15) / 16) <= e->sk_tiles &&
                                gemm_rows_workspace_bytes(M, N, sk) <= e->sk_floats * sizeof(float));
  if (e->dec_fused && M <= DEC_ROWS_MAX && gemm_rows_supported(M, N, K, ln) && fits) {
    GemmArgs g{};
    g.A = A;
    g.lda = lda;
    g.A32 = xln;
    g.lda32 = K;
    g.ln_gamma = ln_g;
    g.ln_beta = ln_b;
    g.ln_eps = 1e-5f;
    g.W = W;
    g.ldw = ldw;
    g.bias = bias;
    g.C = C;
    g.ldc = ldc;
    g.M = M;
    g.N = N;
    g.K = K;
    g.gelu = gelu;
    g.out_mode = out_mode;
    g.site = site;
    g.splitk = sk;
    g.sk_part = e->sk_part[ws];
    g.sk_cnt = e->sk_cnt[ws];
    g.kv_k = kv_k;
    g.kv_v = kv_v;
    g.kv_bs = (long)T_max * K;
    g.kv_t = kv_t;
    g.kv_d = kv_k ? N / 3 : 0;
    HIPCHK(launch_gemm_rows(g, s));
    return WCA_OK;
  }
  if (ln) {
    HIPCHK(launch_layernorm_f16(xln, ln_g, ln_b, xn_scratch, M, K, 1e-5f, s));
    A = xn_scratch;
    lda = K;
  }
  HIPCHK(gemm(s, A, lda, W, ldw, bias, C, ldc, M, N, K, gelu, out_mode, site, e->sk_big[1 + ws], e->sk_big_bytes));
  if (kv_k) HIPCHK(launch_kv_append(reinterpret_cast<const half_t*>(C), kv_k, kv_v, M, T_max, kv_t, N / 3, s));
  return WCA_OK;
}

// x (f32 residual stream, [M][N]) += A W^T + bias, then xn (f16) = LayerNorm(x) with (gamma, beta): ONE kernel where the
// persistent GEMM can exchange the row statistics between the workgroups of a 256-row panel (gemm_epilogue.h, out_mode 3);
// otherwise (few tiles: the decoder, small batches) the read-modify-write GEMM followed by the LayerNorm kernel.
// ev_gemm / ev_ln (profiling): event slots {site, layer} for the GEMM and for the LayerNorm launch; the fused kernel is timed
// as the GEMM's site alone.
int gemm_residual_ln(wca_engine* e, hipStream_t s, const half_t* A, int lda, const half_t* W, int ldw, const float* bias, float* x, int M, int N,
                     int K, const float* gamma, const float* beta, half_t* xn, bool ln_pair, int site, bool allow_fused = true, int ev_gemm_site = -1,
                     int ev_gemm_li = 0, int ev_ln_site = -1, int ev_ln_li = 0, long a_lo = 0, const half_t* w_plain = nullptr, int k_plain = 0) {
  auto ev = [&](int st, int li, int which) {
    if (e->profiling && st >= 0 && li >= 0 && li < 33) {
      (void)hipEventRecord(e->kev[st][li][which], s);
      e->kev_set[st][li] = true;
    }
  };
  ev(ev_gemm_site, ev_gemm_li, 0);
  const int om = ln_pair ? 2 : 1;  // the LayerNorm's consumer is split: xn rows are [hi(N) | lo(N)] (no fused form writes pairs)
  if (allow_fused && !ln_pair && a_lo == 0 && gemm_ln_supported(M, N, K, e->n_cu)) {
    GemmArgs g{};
    g.A = A;
    g.lda = lda;
    g.W = W;
    g.ldw = ldw;
    g.bias = bias;
    g.C = x;
    g.ldc = N;
    g.M = M;
    g.N = N;
    g.K = K;
    g.out_mode = 3;
    g.site = site;
    g.ln_gamma = gamma;
    g.ln_beta = beta;
    g.ln_out = xn;
    g.ln_ld = N;
    g.ln_eps = 1e-5f;
    g.ln_stats = e->ln_stats;
    g.ln_cnt = e->ln_cnt;
    g.ln_err = e->ln_err ? e->ln_err : e->err_dev;
    HIPCHK(launch_gemm(g, s));
    ev(ev_gemm_site, ev_gemm_li, 1);
    return WCA_OK;
  }
  HIPCHK(gemm(s, A, lda, W, ldw, bias, x, N, M, N, K, 0, 2, site, e->sk_big[0], e->sk_big_bytes, 0, a_lo, e, w_plain, k_plain));
  ev(ev_gemm_site, ev_gemm_li, 1);
  ev(ev_ln_site, ev_ln_li, 0);
  HIPCHK(launch_layernorm_f16(x, gamma, beta, xn, M, N, 1e-5f, s, om * N, ln_pair ? N : 0));
  ev(ev_ln_site, ev_ln_li, 1);
  return WCA_OK;
}

// upload helpers: convert host tensor (f32 or f16) into device f16 / f32
// Every weight matrix is f16 AT REST here, like every openai checkpoint ("weights stored fp16, loaded into fp32 params",
// /root/reference/infer_ali.py:36-37), which is what makes A W^T exact-operand arithmetic in the pair mode. An fp32 source whose values are not
// f16-representable (a fine-tuned fp32 state dict) keeps its REMAINDER lo = f16(w - f16(w)) in the W_lo slab (same offset as the f16 value in wslab;
// allocated when the first such tensor arrives), and the pair sites multiply the extra term A_hi W_lo^T (gemm()): w = hi + lo to 2^-22 |w|, the same
// representation the activations travel in. *n_inexact counts the elements with a non-zero remainder (NaN == NaN for this purpose); `base` is the matrix
// the GEMM call sites address (a fused matrix holds several tensors).
int ensure_wlo_slab(wca_engine* e) {
  if (e->wslab_lo) return WCA_OK;
  HIPCHK(hipMalloc((void**)&e->wslab_lo, e->wslab_bytes));
  HIPCHK(hipMemset(e->wslab_lo, 0, e->wslab_bytes));
  return WCA_OK;
}
int put_f16(wca_engine* e, const half_t* base, half_t* dst, const void* src, int dtype, size_t n, size_t* n_inexact) {
  std::vector<half_t> tmp(n);
  size_t bad = 0;
  if (dtype == WCA_DTYPE_F32) {
    const float* s = static_cast<const float*>(src);
    for (size_t i = 0; i < n; ++i) {
      tmp[i] = (half_t)s[i];
      bad += ((float)tmp[i] != s[i]) && (s[i] == s[i]);
    }
    *n_inexact += bad;
  } else {
    memcpy(tmp.data(), src, n * sizeof(half_t));
  }
  HIPCHK(hipMemcpy(dst, tmp.data(), n * sizeof(half_t), hipMemcpyHostToDevice));
  if (bad > 0 || e->wslab_lo) {   // the remainders (zeros when this tensor is exact and replaces an inexact one)
    if (bad > 0) {
      WCA_TRY(ensure_wlo_slab(e));
      const float* s = static_cast<const float*>(src);
      for (size_t i = 0; i < n; ++i) {
        const float r = s[i] - (float)tmp[i];
        tmp[i] = (r == r && std::fabs(r) < 65504.f) ? (half_t)r : (half_t)0.f;
      }
      e->wlo_bases.insert(base);
    } else {
      std::fill(tmp.begin(), tmp.end(), (half_t)0.f);
    }
    HIPCHK(hipMemcpy(e->wslab_lo + (reinterpret_cast<char*>(dst) - e->wslab), tmp.data(), n * sizeof(half_t), hipMemcpyHostToDevice));
  }
  return WCA_OK;
}
int put_f32(float* dst, const void* src, int dtype, size_t n) {
  if (dtype == WCA_DTYPE_F32) {
    HIPCHK(hipMemcpy(dst, src, n * sizeof(float), hipMemcpyHostToDevice));
  } else {
    std::vector<float> tmp(n);
    const half_t* s = static_cast<const half_t*>(src);
    for (size_t i = 0; i < n; ++i) tmp[i] = (float)s[i];
    HIPCHK(hipMemcpy(dst, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice));
  }
  return WCA_OK;
}
inline float host_val(const void* src, int dtype, size_t i) {
  return dtype == WCA_DTYPE_F32 ? static_cast<const float*>(src)[i] : (float)static_cast<const half_t*>(src)[i];
}

// conv weight [out][in][3] -> f16 [out][kpad] with column tap*in + c (remainders of inexact fp32 values into the W_lo slab, like put_f16)
int put_conv(wca_engine* e, half_t* dst, const void* src, int dtype, int out, int in, int kpad, size_t* n_inexact) {
  std::vector<half_t> tmp((size_t)out * kpad, (half_t)0.f), lo((size_t)out * kpad, (half_t)0.f);
  size_t bad = 0;
  for (int n = 0; n < out; ++n)
    for (int c = 0; c < in; ++c)
      for (int t = 0; t < 3; ++t) {
        const float v = host_val(src, dtype, ((size_t)n * in + c) * 3 + t);
        const half_t h = (half_t)v;
        tmp[(size_t)n * kpad + t * in + c] = h;
        if ((float)h != v && v == v) {
          ++bad;
          const float r = v - (float)h;
          lo[(size_t)n * kpad + t * in + c] = std::fabs(r) < 65504.f ? (half_t)r : (half_t)0.f;
        }
      }
  *n_inexact += bad;
  HIPCHK(hipMemcpy(dst, tmp.data(), tmp.size() * sizeof(half_t), hipMemcpyHostToDevice));
  if (bad > 0 || e->wslab_lo) {
    if (bad > 0) {
      WCA_TRY(ensure_wlo_slab(e));
      e->wlo_bases.insert(dst);
    }
    HIPCHK(hipMemcpy(e->wslab_lo + (reinterpret_cast<char*>(dst) - e->wslab), lo.data(), lo.size() * sizeof(half_t), hipMemcpyHostToDevice));
  }
  return WCA_OK;
}

size_t numel(const int64_t* shape, int ndim) {
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  return n;
}

int load_block_tensor(wca_engine* e, LayerW& l, bool is_dec, int li, const std::string& rest, const void* p, int dtype,
                      size_t n, int d, size_t* n_inexact) {
  const size_t dd = (size_t)d * d;
  auto expect = [&](size_t want) -> bool { return n == want; };
#define WANT(cnt) \
  if (!expect(cnt)) return fail(WCA_ERR_INVALID, "weight %s: expected %zu elements, got %zu", rest.c_str(), (size_t)(cnt), n)
  if (rest == "attn.query.weight") { WANT(dd); return put_f16(e, l.qkv_w, l.qkv_w, p, dtype, n, n_inexact); }
  if (rest == "attn.query.bias") { WANT(d); return put_f32(l.qkv_b, p, dtype, n); }
  if (rest == "attn.key.weight") { WANT(dd); return put_f16(e, l.qkv_w, l.qkv_w + dd, p, dtype, n, n_inexact); }
  if (rest == "attn.value.weight") { WANT(dd); return put_f16(e, l.qkv_w, l.qkv_w + 2 * dd, p, dtype, n, n_inexact); }
  if (rest == "attn.value.bias") { WANT(d); return put_f32(l.qkv_b + 2 * d, p, dtype, n); }
  if (rest == "attn.out.weight") { WANT(dd); return put_f16(e, l.out_w, l.out_w, p, dtype, n, n_inexact); }
  if (rest == "attn.out.bias") { WANT(d); return put_f32(l.out_b, p, dtype, n); }
  if (rest == "attn_ln.weight") { WANT(d); return put_f32(l.ln1_g, p, dtype, n); }
  if (rest == "attn_ln.bias") { WANT(d); return put_f32(l.ln1_b, p, dtype, n); }
  if (rest == "mlp.0.weight") { WANT(4 * dd); return put_f16(e, l.fc1_w, l.fc1_w, p, dtype, n, n_inexact); }
  if (rest == "mlp.0.bias") { WANT(4 * (size_t)d); return put_f32(l.fc1_b, p, dtype, n); }
  if (rest == "mlp.2.weight") { WANT(4 * dd); return put_f16(e, l.fc2_w, l.fc2_w, p, dtype, n, n_inexact); }
  if (rest == "mlp.2.bias") { WANT(d); return put_f32(l.fc2_b, p, dtype, n); }
  if (rest == "mlp_ln.weight") { WANT(d); return put_f32(l.ln2_g, p, dtype, n); }
  if (rest == "mlp_ln.bias") { WANT(d); return put_f32(l.ln2_b, p, dtype, n); }
  if (is_dec) {
    if (rest == "cross_attn.query.weight") { WANT(dd); return put_f16(e, l.cq_w, l.cq_w, p, dtype, n, n_inexact); }
    if (rest == "cross_attn.query.bias") { WANT(d); return put_f32(l.cq_b, p, dtype, n); }
    if (rest == "cross_attn.key.weight") { WANT(dd); return put_f16(e, e->kv_w, e->kv_w + (size_t)(2 * li) * dd, p, dtype, n, n_inexact); }
    if (rest == "cross_attn.value.weight") { WANT(dd); return put_f16(e, e->kv_w, e->kv_w + (size_t)(2 * li + 1) * dd, p, dtype, n, n_inexact); }
    if (rest == "cross_attn.value.bias") { WANT(d); return put_f32(e->kv_b + (size_t)(2 * li + 1) * d, p, dtype, n); }
    if (rest == "cross_attn.out.weight") { WANT(dd); return put_f16(e, l.co_w, l.co_w, p, dtype, n, n_inexact); }
    if (rest == "cross_attn.out.bias") { WANT(d); return put_f32(l.co_b, p, dtype, n); }
    if (rest == "cross_attn_ln.weight") { WANT(d); return put_f32(l.lnc_g, p, dtype, n); }
    if (rest == "cross_attn_ln.bias") { WANT(d); return put_f32(l.lnc_b, p, dtype, n); }
  }
#undef WANT
  return 1;  // unknown (ignored)
}

void record(wca_engine* e, int i, hipStream_t s = nullptr) {
  if (e->profiling && e->ev_valid) (void)hipEventRecord(e->ev[i], s ? s : e->stream);
}

// Entry points other than wca_align_batch* share the phase-2 scratch (capture, norms, DTW buffers) with a batch
// that may still be running on stream2: order them behind it.
int join_phase2(wca_engine* e) {
  if (e->enq_count > e->fetch_count) HIPCHK(hipStreamWaitEvent(e->stream, e->res_ev[(e->enq_count - 1) & 1], 0));
  return WCA_OK;
}

// ---- encoder: mel_tm (f16 time-major) -> xn = ln_post(x) (f16) and optionally x (f32)
// Per-site precision (e->sites): a split stage runs the same launches on [hi | lo] operand rows (row width 2 * width, lo half
// `width` elements after the hi half) against the K-doubled weight copies; a producer stores pairs (out_mode 4 / the LayerNorm's
// lo_off) exactly when its consumer is split; the fp32 residual stream is the same in every mode.
int run_encoder(wca_engine* e, int B) {
  const wca_model_dims& D = e->dims;
  const int d = D.n_audio_state, H = D.n_audio_head;
  const bool cv = site_on(e, WCA_PSITE_CONV);
  const int omc = cv ? 2 : 1;
  hipStream_t s = e->stream;
  {
    GemmArgs g{};
    g.A = e->mel_tm;
    g.lda = omc * D.n_mels;
    g.a_rows_per_batch = N_FRAMES;
    g.a_batch_stride = (long)(N_FRAMES + 2) * omc * D.n_mels;
    g.W = cv ? e->sw.conv1_w : e->conv1_w;
    g.ldw = cv ? e->sw.k1pad : e->k1pad;
    g.bias = e->conv1_b;
    g.C = e->h1pad + omc * d;  // output frame t lands in padded row t + 1
    g.ldc = omc * d;
    g.c_rows_per_batch = N_FRAMES;
    g.c_batch_stride = (long)(N_FRAMES + 2) * omc * d;
    g.c_lo = cv ? d : 0;
    g.M = B * N_FRAMES;
    g.N = d;
    g.K = g.ldw;
    g.gelu = 1;
    g.out_mode = cv ? 4 : 0;
    g.site = 3;
    if (cv && use_wlo(e) && e->wlo_bases.count(e->conv1_w)) {   // inexact conv1 weights: the A_hi W_lo^T term, added before the GELU
      GemmArgs x = g;
      x.W = e->sw.conv1_wlo;
      x.bias = nullptr;
      x.gelu = 0;
      x.out_mode = 1;
      x.c_lo = 0;
      x.c_rows_per_batch = 0;
      HIPCHK(e->wlo_tmp[0].ensure((size_t)g.M * g.N * sizeof(float)));
      x.C = e->wlo_tmp[0].p;
      x.ldc = g.N;
      HIPCHK(launch_gemm(x, s));
      g.addend = (const float*)e->wlo_tmp[0].p;
      g.ld_addend = g.N;
    }
    HIPCHK(launch_gemm(g, s));
  }
  {
    GemmArgs g{};
    g.A = e->h1pad;
    g.lda = 2 * omc * d;  // stride 2
    g.a_rows_per_batch = N_CTX;
    g.a_batch_stride = (long)(N_FRAMES + 2) * omc * d;
    g.W = cv ? e->sw.conv2_w : e->conv2_w;
    g.ldw = 3 * omc * d;
    g.bias = e->conv2_b;
    g.C = e->x;
    g.ldc = d;
    g.pos = e->enc_pos;
    g.pos_period = N_CTX;
    g.M = B * N_CTX;
    g.N = d;
    g.K = 3 * omc * d;
    g.gelu = 1;
    g.out_mode = 1;
    g.site = 3;
    if (cv && use_wlo(e) && e->wlo_bases.count(e->conv2_w)) {
      GemmArgs x = g;
      x.W = e->sw.conv2_wlo;
      x.bias = nullptr;
      x.pos = nullptr;
      x.gelu = 0;
      x.out_mode = 1;
      HIPCHK(e->wlo_tmp[0].ensure((size_t)g.M * g.N * sizeof(float)));
      x.C = e->wlo_tmp[0].p;
      x.ldc = g.N;
      HIPCHK(launch_gemm(x, s));
      g.addend = (const float*)e->wlo_tmp[0].p;
      g.ld_addend = g.N;
    }
    HIPCHK(launch_gemm(g, s));
  }
  const int M = B * N_CTX;
  const float scale = 1.0f / std::sqrt((float)(d / H));
  memset(e->kev_set, 0, sizeof(e->kev_set));
  auto mark = [&](int site, int li, int which) {
    if (e->profiling && li < 33) {
      (void)hipEventRecord(e->kev[site][li][which], s);
      e->kev_set[site][li] = true;
    }
  };
  // LayerNorms ride in the epilogue of the GEMM that produces their input (gemm_residual_ln) where wca_set_fuse_ln allows it and
  // their consumer reads single f16 rows: mlp_ln in the attention out-projection, the NEXT layer's attn_ln (ln_post after the
  // last layer) in fc2; only layer 0's attn_ln is always a launch
  {
    const bool p0 = enc_gemm_split(e, 0);
    mark(WCA_SITE_LN1, 0, 0);
    HIPCHK(launch_layernorm_f16(e->x, e->enc[0].ln1_g, e->enc[0].ln1_b, e->xn, M, d, 1e-5f, s, (p0 ? 2 : 1) * d, p0 ? d : 0));
    mark(WCA_SITE_LN1, 0, 1);
  }
  for (int li = 0; li < D.n_audio_layer; ++li) {
    const LayerW& l = e->enc[li];
    const bool gs = enc_gemm_split(e, li), as = enc_attn_split(e, li);
    const LayerW& w2 = e->split ? e->sw.enc[li] : l;  // the K-doubled copies [N][2K] = [W | W] (present while any site is split)
    const int oma = as ? 2 : 1;  // q / k / v rows and the attention output: pairs iff the attention is split
    // q / k / v projection: xn is a pair buffer iff this layer's GEMMs are split (its LayerNorm wrote it for them)
    const GemmOpnd oq = pick_operands(gs, gs, l.qkv_w, w2.qkv_w, d, M, 3 * d, as ? 4 : 0, e);
    mark(WCA_SITE_QKV, li, 0);
    HIPCHK(gemm(s, e->xn, oq.lda, oq.W, oq.ldw, l.qkv_b, e->qkv, oma * 3 * d, M, 3 * d, oq.K, 0, 0, 1, nullptr, 0, as ? 3 * d : 0, oq.a_lo, e, oq.Wp, oq.Kp));
    mark(WCA_SITE_QKV, li, 1);
    AttnArgs a{};
    a.Q = e->qkv;
    a.K = e->qkv + d;
    a.V = e->qkv + 2 * d;
    a.q_bs = a.k_bs = a.v_bs = (long)N_CTX * oma * 3 * d;
    a.q_rs = a.k_rs = a.v_rs = oma * 3 * d;
    a.O = e->att;
    a.o_bs = (long)N_CTX * oma * d;
    a.o_rs = oma * d;
    a.split = as ? 1 : 0;
    a.q_lo = a.k_lo = a.v_lo = 3 * d;
    a.o_lo = d;
    a.nq = N_CTX;
    a.nk = N_CTX;
    a.H = H;
    a.B = B;
    a.scale = scale;
    a.causal = 0;
    mark(WCA_SITE_ATTN, li, 0);
    HIPCHK(launch_attention(a, s));
    mark(WCA_SITE_ATTN, li, 1);
    // sites OUT / FC2 = the GEMM alone (or the fused GEMM + LayerNorm kernel); the LayerNorm launches: mlp_ln = LN2[li], the next
    // layer's attn_ln / ln_post = LN1[li + 1]
    const GemmOpnd oo = pick_operands(as, gs, l.out_w, w2.out_w, d, M, d, 2, e);
    if (int rc = gemm_residual_ln(e, s, e->att, oo.lda, oo.W, oo.ldw, l.out_b, e->x, M, d, oo.K, l.ln2_g, l.ln2_b, e->xn, gs, 1, e->fuse_ln, WCA_SITE_OUT, li,
                                  WCA_SITE_LN2, li, oo.a_lo, oo.Wp, oo.Kp))
      return rc;
    const GemmOpnd o1 = pick_operands(gs, gs, l.fc1_w, w2.fc1_w, d, M, 4 * d, gs ? 4 : 0, e);
    mark(WCA_SITE_FC1, li, 0);
    HIPCHK(gemm(s, e->xn, o1.lda, o1.W, o1.ldw, l.fc1_b, e->hid, (gs ? 2 : 1) * 4 * d, M, 4 * d, o1.K, 1, 0, 1, nullptr, 0, gs ? 4 * d : 0, o1.a_lo, e, o1.Wp, o1.Kp));
    mark(WCA_SITE_FC1, li, 1);
    const bool last = li + 1 == D.n_audio_layer;
    // the LayerNorm behind fc2 feeds the next layer's q / k / v projection, or (ln_post) the cross-K/V projection
    const bool next_pair = last ? site_on(e, WCA_PSITE_CROSS_KV) : enc_gemm_split(e, li + 1);
    const GemmOpnd o2 = pick_operands(gs, gs, l.fc2_w, w2.fc2_w, 4 * d, M, d, 2, e);
    if (int rc = gemm_residual_ln(e, s, e->hid, o2.lda, o2.W, o2.ldw, l.fc2_b, e->x, M, d, o2.K, last ? e->lnpost_g : e->enc[li + 1].ln1_g,
                                  last ? e->lnpost_b : e->enc[li + 1].ln1_b, e->xn, next_pair, 4, e->fuse_ln, WCA_SITE_FC2, li, WCA_SITE_LN1, li + 1, o2.a_lo, o2.Wp, o2.Kp))
      return rc;
  }
  return WCA_OK;
}

// cross-attention K/V of every decoder layer in one GEMM: kv[b*1500 + t][(2l + {0,1})*dt + c]
// skip_last_v: the value projection of the LAST decoder layer (the final dt columns) is only read by that layer's
// P.V product, whose result nobody uses when the caller wants the captured logits but no output logits.
// The rows are pairs [hi(L*2*dt) | lo(L*2*dt)] iff the hooked cross-attention (CAPTURE) is split.
int run_cross_kv(wca_engine* e, int B, half_t* kvbuf = nullptr, bool skip_last_v = false) {
  if (!kvbuf) kvbuf = e->kv;
  const wca_model_dims& D = e->dims;
  const int d = D.n_audio_state, dt = D.n_text_state, L = D.n_text_layer;
  const int n_cols = L * 2 * dt - (skip_last_v ? dt : 0);
  const bool ks = site_on(e, WCA_PSITE_CROSS_KV), cs = site_on(e, WCA_PSITE_CAPTURE);
  const GemmOpnd o = pick_operands(ks, ks, e->kv_w, e->split ? e->sw.kv_w : e->kv_w, d, B * N_CTX, n_cols, cs ? 4 : 0, e);
  HIPCHK(gemm(e->stream, e->xn, o.lda, o.W, o.ldw, e->kv_b, kvbuf, (cs ? 2 : 1) * L * 2 * dt, B * N_CTX, n_cols, o.K, 0, 0, 3, nullptr, 0,
              cs ? (long)L * 2 * dt : 0, o.a_lo, e, o.Wp, o.Kp));
  return WCA_OK;
}

// The teacher-forced decoder with a split site (DEC: its LayerNorms / GEMMs / causal self-attention on pairs; CAPTURE: the hooked
// cross-attention on q / K / V pairs): separate LayerNorm launches, the tile GEMMs (the few-row kernel of gemm_rows.hip has no pair
// output), attn_split_kernel where the attention is split. The captured logits of a split CAPTURE are the three-pass fp32 sums.
int run_decoder_sites(wca_engine* e, const int64_t* tokens_dev, int B, int n, float* cap, int Fpad, int Fcap, float* logits_out, hipStream_t s,
                      const half_t* kvbuf) {
  const wca_model_dims& D = e->dims;
  const int dt = D.n_text_state, H = D.n_text_head, L = D.n_text_layer;
  const int M = B * n;
  const float scale = 1.0f / std::sqrt((float)(dt / H));
  const bool gs = site_on(e, WCA_PSITE_DEC), cs = site_on(e, WCA_PSITE_CAPTURE);
  const int omg = gs ? 2 : 1, omx = cs ? 2 : 1;
  const int kv_ld = omx * L * 2 * dt;
  const long kv_lo = (long)L * 2 * dt;
  HIPCHK(launch_embed(tokens_dev, e->tok_emb, e->dec_pos, e->xd, B, n, dt, D.n_vocab, e->err_dev, s,
                      (gs && use_wlo(e) && e->wlo_bases.count(e->tok_emb)) ? wlo_of(e, e->tok_emb) : nullptr));
  auto ln = [&](const float* g, const float* b) -> int {
    HIPCHK(launch_layernorm_f16(e->xd, g, b, e->xdn, M, dt, 1e-5f, s, omg * dt, gs ? dt : 0));
    return WCA_OK;
  };
  // C = A W^T (+ bias ...): a_pair = the A buffer holds [hi | lo] rows of K values each; c_lo > 0: f16 pair output
  auto mm = [&](const half_t* A, bool a_pair, const half_t* W1, const half_t* W2, const float* bias, void* C, int ldc, int N, int K, int gelu, int out_mode,
                long c_lo, int site) -> int {
    const GemmOpnd o = pick_operands(a_pair, gs, W1, W2, K, M, N, (c_lo > 0 && out_mode == 0) ? 4 : out_mode, e);
    HIPCHK(gemm(s, A, o.lda, o.W, o.ldw, bias, C, ldc, M, N, o.K, gelu, out_mode, site, e->sk_big[1], e->sk_big_bytes, c_lo, o.a_lo, e, o.Wp, o.Kp));
    return WCA_OK;
  };
  for (int li = 0; li < L; ++li) {
    const LayerW& l = e->dec[li];
    const LayerW& w = e->sw.dec[li];
    WCA_TRY(ln(l.ln1_g, l.ln1_b));
    WCA_TRY(mm(e->xdn, gs, l.qkv_w, w.qkv_w, l.qkv_b, e->qkv_d, omg * 3 * dt, 3 * dt, dt, 0, 0, gs ? 3 * dt : 0, 2));
    {
      AttnArgs a{};
      a.Q = e->qkv_d;
      a.K = e->qkv_d + dt;
      a.V = e->qkv_d + 2 * dt;
      a.q_bs = a.k_bs = a.v_bs = (long)n * omg * 3 * dt;
      a.q_rs = a.k_rs = a.v_rs = omg * 3 * dt;
      a.O = e->att_d;
      a.o_bs = (long)n * omg * dt;
      a.o_rs = omg * dt;
      a.split = gs ? 1 : 0;
      a.q_lo = a.k_lo = a.v_lo = 3 * dt;
      a.o_lo = dt;
      a.nq = n;
      a.nk = n;
      a.H = H;
      a.B = B;
      a.scale = scale;
      a.causal = 1;
      HIPCHK(launch_attention(a, s));
    }
    WCA_TRY(mm(e->att_d, gs, l.out_w, w.out_w, l.out_b, e->xd, dt, dt, dt, 0, 2, 0, 2));
    WCA_TRY(ln(l.lnc_g, l.lnc_b));
    WCA_TRY(mm(e->xdn, gs, l.cq_w, w.cq_w, l.cq_b, e->q_d, omx * dt, dt, dt, 0, 0, cs ? dt : 0, 2));
    {
      AttnArgs a{};
      a.Q = e->q_d;
      a.q_bs = (long)n * omx * dt;
      a.q_rs = omx * dt;
      a.K = kvbuf + (size_t)(2 * li) * dt;
      a.V = kvbuf + (size_t)(2 * li + 1) * dt;
      a.k_bs = a.v_bs = (long)N_CTX * kv_ld;
      a.k_rs = a.v_rs = kv_ld;
      a.O = e->att_d;
      a.o_bs = (long)n * omx * dt;
      a.o_rs = omx * dt;
      a.split = cs ? 1 : 0;
      a.q_lo = dt;
      a.k_lo = a.v_lo = kv_lo;
      a.o_lo = dt;
      a.cap = cap ? cap + (size_t)li * H * n * Fpad : nullptr;
      a.cap_bs = (long)L * H * n * Fpad;
      a.cap_hs = (long)n * Fpad;
      a.cap_ld = Fpad;
      a.cap_cols = Fcap;
      a.nq = n;
      a.nk = N_CTX;
      a.H = H;
      a.B = B;
      a.scale = scale;
      a.causal = 0;
      HIPCHK(launch_attention(a, s));
    }
    if (li == L - 1 && !logits_out) break;
    WCA_TRY(mm(e->att_d, cs, l.co_w, w.co_w, l.co_b, e->xd, dt, dt, dt, 0, 2, 0, 2));
    WCA_TRY(ln(l.ln2_g, l.ln2_b));
    WCA_TRY(mm(e->xdn, gs, l.fc1_w, w.fc1_w, l.fc1_b, e->hid_d, omg * 4 * dt, 4 * dt, dt, 1, 0, gs ? 4 * dt : 0, 2));
    WCA_TRY(mm(e->hid_d, gs, l.fc2_w, w.fc2_w, l.fc2_b, e->xd, dt, dt, 4 * dt, 0, 2, 0, 2));
  }
  if (logits_out) {
    WCA_TRY(ln(e->lnf_g, e->lnf_b));
    WCA_TRY(mm(e->xdn, gs, e->tok_emb, e->sw.tok_emb, nullptr, logits_out, D.n_vocab, D.n_vocab, dt, 0, 1, 0, 3));
  }
  return WCA_OK;
}

// decoder with capture. tokens_dev [B][n]; capture -> cap [B][L*H][n][Fpad] (first Fcap keys)
int run_decoder(wca_engine* e, const int64_t* tokens_dev, int B, int n, float* cap, int Fpad, int Fcap, float* logits_out,
                hipStream_t s = nullptr, const half_t* kvbuf = nullptr) {
  const wca_model_dims& D = e->dims;
  const int dt = D.n_text_state, H = D.n_text_head, L = D.n_text_layer;
  if (!s) s = e->stream;
  if (!kvbuf) kvbuf = e->kv;
  if (site_on(e, WCA_PSITE_DEC) || site_on(e, WCA_PSITE_CAPTURE)) return run_decoder_sites(e, tokens_dev, B, n, cap, Fpad, Fcap, logits_out, s, kvbuf);
  const int M = B * n;
  const float scale = 1.0f / std::sqrt((float)(dt / H));
  HIPCHK(launch_embed(tokens_dev, e->tok_emb, e->dec_pos, e->xd, B, n, dt, D.n_vocab, e->err_dev, s));
  for (int li = 0; li < L; ++li) {
    const LayerW& l = e->dec[li];
    WCA_TRY(dec_gemm(e, s, 0, nullptr, 0, e->xd, l.ln1_g, l.ln1_b, e->xdn, l.qkv_w, dt, l.qkv_b, e->qkv_d, 3 * dt, M, 3 * dt, dt, 0, 0, 2));
    {
      AttnArgs a{};
      a.Q = e->qkv_d;
      a.K = e->qkv_d + dt;
      a.V = e->qkv_d + 2 * dt;
      a.q_bs = a.k_bs = a.v_bs = (long)n * 3 * dt;
      a.q_rs = a.k_rs = a.v_rs = 3 * dt;
      a.O = e->att_d;
      a.o_bs = (long)n * dt;
      a.o_rs = dt;
      a.nq = n;
      a.nk = n;
      a.H = H;
      a.B = B;
      a.scale = scale;
      a.causal = 1;
      HIPCHK(launch_attention(a, s));
    }
    WCA_TRY(dec_gemm(e, s, 0, e->att_d, dt, nullptr, nullptr, nullptr, nullptr, l.out_w, dt, l.out_b, e->xd, dt, M, dt, dt, 0, 2, 2));
    WCA_TRY(dec_gemm(e, s, 0, nullptr, 0, e->xd, l.lnc_g, l.lnc_b, e->xdn, l.cq_w, dt, l.cq_b, e->q_d, dt, M, dt, dt, 0, 0, 2));
    {
      AttnArgs a{};
      a.Q = e->q_d;
      a.q_bs = (long)n * dt;
      a.q_rs = dt;
      a.K = kvbuf + (size_t)(2 * li) * dt;
      a.V = kvbuf + (size_t)(2 * li + 1) * dt;
      a.k_bs = a.v_bs = (long)N_CTX * L * 2 * dt;
      a.k_rs = a.v_rs = L * 2 * dt;
      a.O = e->att_d;
      a.o_bs = (long)n * dt;
      a.o_rs = dt;
      a.cap = cap ? cap + (size_t)li * H * n * Fpad : nullptr;
      a.cap_bs = (long)L * H * n * Fpad;
      a.cap_hs = (long)n * Fpad;
      a.cap_ld = Fpad;
      a.cap_cols = Fcap;
      a.nq = n;
      a.nk = N_CTX;
      a.H = H;
      a.B = B;
      a.scale = scale;
      a.causal = 0;
      HIPCHK(launch_attention(a, s));
    }
    // the last layer's cross-attention logits are captured by now: without logits nothing downstream is read
    if (li == L - 1 && !logits_out) break;
    WCA_TRY(dec_gemm(e, s, 0, e->att_d, dt, nullptr, nullptr, nullptr, nullptr, l.co_w, dt, l.co_b, e->xd, dt, M, dt, dt, 0, 2, 2));
    WCA_TRY(dec_gemm(e, s, 0, nullptr, 0, e->xd, l.ln2_g, l.ln2_b, e->xdn, l.fc1_w, dt, l.fc1_b, e->hid_d, 4 * dt, M, 4 * dt, dt, 1, 0, 2));
    WCA_TRY(dec_gemm(e, s, 0, e->hid_d, 4 * dt, nullptr, nullptr, nullptr, nullptr, l.fc2_w, 4 * dt, l.fc2_b, e->xd, dt, M, dt, 4 * dt, 0, 2, 2));
  }
  if (logits_out) {
    WCA_TRY(dec_gemm(e, s, 0, nullptr, 0, e->xd, e->lnf_g, e->lnf_b, e->xdn, e->tok_emb, dt, nullptr, logits_out, D.n_vocab, M, D.n_vocab, dt, 0, 1, 3));
  }
  return WCA_OK;
}

// One autoregressive step of the greedy ASR pre-pass for rows [b0, b0 + B) of the batch: position t of every row (token
// tokens[b][t]) through the decoder with the self-attention K/V cache (positions 0..t), cross-attention over this batch's
// cross-K/V; logits of that position -> e->dec_logits. `ws` = which split-K workspace (one per decode stream). Eight
// launches per layer: [LN1 + QKV + cache append], self-attention, [out-projection + residual], [LNc + cross query],
// cross-attention, [cross out + residual], [LN2 + fc1 + GELU], [fc2 + residual, split-K].
// phase: -1 = embedding only, li in [0, L) = decoder layer li only, L = final LayerNorm + logits only, -2 = the whole step.
// The two half-batches of wca_greedy_decode are enqueued layer by layer in turn (the queues are served in the order their
// packets arrive: coarse enqueueing gives coarse alternation and no overlap).
int run_decode_step(wca_engine* e, hipStream_t s, int ws, const half_t* kvbuf, const int* tokens, int b0, int B, int B_all, int t, int T_max,
                    bool want_logits, int phase = -2) {
  const wca_model_dims& D = e->dims;
  const int dt = D.n_text_state, H = D.n_text_head, L = D.n_text_layer;
  const float scale = 1.0f / std::sqrt((float)(dt / H));
  half_t* cache = (half_t*)e->dec_cache.p;
  const size_t plane = (size_t)B_all * T_max * dt;  // one layer's K (or V) cache
  float* xd = e->xd + (size_t)b0 * dt;
  half_t* xdn = e->xdn + (size_t)b0 * dt;
  half_t* qkv_d = e->qkv_d + (size_t)b0 * 3 * dt;
  half_t* att_d = e->att_d + (size_t)b0 * dt;
  half_t* q_d = e->q_d + (size_t)b0 * dt;
  half_t* hid_d = e->hid_d + (size_t)b0 * 4 * dt;
  // split mode: the cross-K/V rows are [hi | lo]; the greedy pre-pass (whisper.decode runs in fp16 itself) reads the hi halves
  const int kv_ld = (site_on(e, WCA_PSITE_CAPTURE) ? 2 : 1) * L * 2 * dt;
  const half_t* kvb = kvbuf + (size_t)b0 * N_CTX * kv_ld;
  if (phase == -2 || phase == -1)
    HIPCHK(launch_embed_step(tokens + (size_t)b0 * T_max, T_max, t, e->tok_emb, e->dec_pos, xd, B, dt, D.n_vocab, s));
  for (int li = 0; li < L; ++li) {
    if (phase != -2 && phase != li) continue;
    const LayerW& l = e->dec[li];
    half_t* kc = cache + (size_t)(2 * li) * plane + (size_t)b0 * T_max * dt;
    half_t* vc = kc + plane;
    WCA_TRY(dec_gemm(e, s, ws, nullptr, 0, xd, l.ln1_g, l.ln1_b, xdn, l.qkv_w, dt, l.qkv_b, qkv_d, 3 * dt, B, 3 * dt, dt, 0, 0, 2, kc, vc, T_max, t));
    {
      AttnArgs a{};
      a.Q = qkv_d;
      a.q_bs = 3 * dt;
      a.q_rs = 3 * dt;
      a.K = kc;
      a.V = vc;
      a.k_bs = a.v_bs = (long)T_max * dt;
      a.k_rs = a.v_rs = dt;
      a.O = att_d;
      a.o_bs = dt;
      a.o_rs = dt;
      a.nq = 1;
      a.nk = t + 1;  // the cache holds exactly the causal prefix
      a.H = H;
      a.B = B;
      a.scale = scale;
      a.causal = 0;
      HIPCHK(launch_attention(a, s));
    }
    WCA_TRY(dec_gemm(e, s, ws, att_d, dt, nullptr, nullptr, nullptr, nullptr, l.out_w, dt, l.out_b, xd, dt, B, dt, dt, 0, 2, 2));
    WCA_TRY(dec_gemm(e, s, ws, nullptr, 0, xd, l.lnc_g, l.lnc_b, xdn, l.cq_w, dt, l.cq_b, q_d, dt, B, dt, dt, 0, 0, 2));
    {
      AttnArgs a{};
      a.Q = q_d;
      a.q_bs = dt;
      a.q_rs = dt;
      a.K = kvb + (size_t)(2 * li) * dt;
      a.V = kvb + (size_t)(2 * li + 1) * dt;
      a.k_bs = a.v_bs = (long)N_CTX * kv_ld;
      a.k_rs = a.v_rs = kv_ld;
      a.O = att_d;
      a.o_bs = dt;
      a.o_rs = dt;
      a.nq = 1;
      a.nk = N_CTX;
      a.H = H;
      a.B = B;
      a.scale = scale;
      a.causal = 0;
      HIPCHK(launch_attention(a, s));
    }
    WCA_TRY(dec_gemm(e, s, ws, att_d, dt, nullptr, nullptr, nullptr, nullptr, l.co_w, dt, l.co_b, xd, dt, B, dt, dt, 0, 2, 2));
    WCA_TRY(dec_gemm(e, s, ws, nullptr, 0, xd, l.ln2_g, l.ln2_b, xdn, l.fc1_w, dt, l.fc1_b, hid_d, 4 * dt, B, 4 * dt, dt, 1, 0, 2));
    WCA_TRY(dec_gemm(e, s, ws, hid_d, 4 * dt, nullptr, nullptr, nullptr, nullptr, l.fc2_w, 4 * dt, l.fc2_b, xd, dt, B, dt, 4 * dt, 0, 2, 2));
  }
  if (want_logits && (phase == -2 || phase == L))
    WCA_TRY(dec_gemm(e, s, ws, nullptr, 0, xd, e->lnf_g, e->lnf_b, xdn, e->tok_emb, dt, nullptr, (float*)e->dec_logits.p + (size_t)b0 * D.n_vocab,
                     D.n_vocab, B, D.n_vocab, dt, 0, 1, 3));
  return WCA_OK;
}

int check_ready(wca_engine* e) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  if (!e->finalized) return fail(WCA_ERR_STATE, "weights not finalized (call wca_finalize_weights)");
  HIPCHK(hipSetDevice(e->device));
  return ensure_split_weights(e);  // split mode: the K-doubled weight copies are current (no-op otherwise)
}

// stage per-utterance metadata into the next device slot: rows = {n_samples, n_tok, n_frames, dtwN}
int stage_meta(wca_engine* e, int B, const int32_t* a0, const int32_t* a1, const int32_t* a2, const int32_t* a3, int** dev_rows,
               hipStream_t s = nullptr) {
  const int slot = e->meta_slot;
  e->meta_slot = (e->meta_slot + 1) % META_SLOTS;
  int* h = e->meta_host + (size_t)slot * 4 * e->max_batch;
  int* dv = e->meta_dev + (size_t)slot * 4 * e->max_batch;
  const int32_t* src[4] = {a0, a1, a2, a3};
  for (int r = 0; r < 4; ++r)
    for (int b = 0; b < B; ++b) h[r * e->max_batch + b] = src[r] ? src[r][b] : 0;
  HIPCHK(hipMemcpyAsync(dv, h, sizeof(int) * 4 * e->max_batch, hipMemcpyHostToDevice, s ? s : e->stream));
  for (int r = 0; r < 4; ++r) dev_rows[r] = dv + r * e->max_batch;
  return WCA_OK;
}

int run_logmel(wca_engine* e, const float* pcm_dev, int64_t pcm_stride, const int* n_samples_dev, int B, float* mel_out, bool want_tm) {
  if (!e->have_filters) return fail(WCA_ERR_STATE, "mel_filters not loaded (wca_load_weight(\"mel_filters\"))");
  LogMelArgs a{};
  a.pcm = pcm_dev;
  a.pcm_stride = pcm_stride;
  a.n_samples = n_samples_dev;
  a.filters = e->mel_filters;
  a.filt_lo = e->filt_lo;
  a.filt_hi = e->filt_hi;
  a.window = e->window;
  a.twiddle = e->twiddle;
  a.mel_out = mel_out;
  a.mel_tm = want_tm ? e->mel_tm : nullptr;
  const bool cv = site_on(e, WCA_PSITE_CONV);  // the conv stem reads pairs
  a.n_mels_pad = (cv ? 2 : 1) * e->dims.n_mels;
  a.tm_lo = cv ? e->dims.n_mels : 0;
  a.precise = site_on(e, WCA_PSITE_LOGMEL) ? 1 : 0;
  a.scratch = e->mel_scratch;
  a.gmax = e->gmax;
  a.n_mels = e->dims.n_mels;
  a.B = B;
  HIPCHK(launch_logmel(a, e->stream));
  return WCA_OK;
}

// mel f32 [B][n_mels][3000] -> time-major f16 image used by the conv GEMM
// row = row length of the image (n_mels, or 2 n_mels in split mode: lo = f16(v - hi) at column lo_off + m)
__global__ void mel_to_tm_kernel(const float* __restrict__ mel, half_t* __restrict__ tm, int n_mels, int B, int row, int lo_off) {
  const int b = blockIdx.y;
  const long e0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e0 >= (long)n_mels * N_FRAMES) return;
  const int t = (int)(e0 / n_mels), m = (int)(e0 - (long)t * n_mels);
  const float v = mel[((long)b * n_mels + m) * N_FRAMES + t];
  const half_t hv = (half_t)v;
  half_t* o = tm + ((long)b * (N_FRAMES + 2) + t + 1) * row + m;
  o[0] = hv;
  if (lo_off) o[lo_off] = (half_t)(v - (float)hv);  // (v is a loaded value: nothing to contract into either conversion)
}

int mel_to_tm(wca_engine* e, const float* mel_dev, int batch) {
  const wca_model_dims& D = e->dims;
  const size_t nel = (size_t)D.n_mels * N_FRAMES;
  dim3 grid((unsigned)((nel + 255) / 256), batch);
  const bool cv = site_on(e, WCA_PSITE_CONV);
  hipLaunchKernelGGL(mel_to_tm_kernel, grid, dim3(256), 0, e->stream, mel_dev, e->mel_tm, D.n_mels, batch, (cv ? 2 : 1) * D.n_mels, cv ? D.n_mels : 0);
  HIPCHK(hipGetLastError());
  return WCA_OK;
}

// out[r][c] = hi + lo of a split row [hi(d) | lo(d)]
__global__ void widen_split_kernel(const half_t* __restrict__ in, float* __restrict__ out, size_t rows, int d) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < rows * d; i += st) {
    const size_t r = i / d, c = i - r * d;
    out[i] = (float)in[r * 2 * d + c] + (float)in[r * 2 * d + d + c];
  }
}

__global__ void widen_kernel(const half_t* __restrict__ in, float* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) out[i] = (float)in[i];
}

// K/V slot for a new encode: a free one, else the slot of the oldest state that was decoded but never aligned
// (a stand-alone wca_greedy_decode); -1 if both hold data that is still needed.
int take_kv_slot(wca_engine* e) {
  for (int sl = 0; sl < 2; ++sl)
    if (!e->slot_busy[sl]) return sl;
  for (auto it = e->enc_q.begin(); it != e->enc_q.end(); ++it)
    if (it->decoded) {
      const int sl = it->slot;
      e->enc_q.erase(it);
      return sl;
    }
  return -1;
}

// Phase 1 on `stream` for one micro-batch: log-mel (from PCM) or layout change (from a given mel), encoder, cross-K/V of
// every decoder layer into K/V slot `slot`; records ev_kv[slot]. n_samples_dev is only needed with pcm_dev.
int run_phase1(wca_engine* e, const float* mel_dev, const float* pcm_dev, int64_t pcm_stride, const int* n_samples_dev, int batch, int slot,
               bool skip_last_v = false) {
  const wca_model_dims& D = e->dims;
  half_t* kvbuf = slot ? e->kv_alt : e->kv;
  record(e, 0);
  if (pcm_dev) {
    int rc = run_logmel(e, pcm_dev, pcm_stride, n_samples_dev, batch, nullptr, true);
    if (rc) return rc;
  } else {
    if (int mr = mel_to_tm(e, mel_dev, batch)) return mr;
  }
  record(e, 1);
  e->ln_err = e->err_dev + 1 + slot;
  HIPCHK(hipMemsetAsync(e->ln_err, 0, sizeof(int), e->stream));
  g_gemm_cu_limit = e->part_cus > 0 ? e->n_cu - e->part_cus : 0;   // persistent GEMM grids = the CUs phase 1's stream owns
  int rc = run_encoder(e, batch);
  e->ln_err = e->err_dev;
  if (!rc) {
    record(e, 2);
    rc = run_cross_kv(e, batch, kvbuf, skip_last_v);
  }
  g_gemm_cu_limit = 0;
  if (rc) return rc;
  record(e, 3);
  HIPCHK(hipEventRecord(e->ev_kv[slot], e->stream));
  return WCA_OK;
}

int check_pcm_lengths(const int32_t* n_samples_host, int batch, int64_t pcm_stride) {
  for (int b = 0; b < batch; ++b)
    if (n_samples_host[b] < 0 || n_samples_host[b] > 480000 || n_samples_host[b] > pcm_stride)
      return fail(WCA_ERR_INVALID, "n_samples[%d]=%d invalid (pad_or_trim to <= 480000 first)", b, n_samples_host[b]);
  return WCA_OK;
}

int validate_lengths(int B, int n_tok_max, const int32_t* n_tok, const int32_t* max_frames, int* Fmax_out) {
  if (n_tok_max > MAX_TOK) return fail(WCA_ERR_TOO_LONG, "n_tok %d > %d", n_tok_max, MAX_TOK);
  if (n_tok_max < 1) return fail(WCA_ERR_INVALID, "n_tok %d < 1", n_tok_max);
  int Fmax = 0;
  for (int b = 0; b < B; ++b) {
    if (n_tok && (n_tok[b] > n_tok_max || n_tok[b] < 0)) return fail(WCA_ERR_INVALID, "n_tok[%d]=%d outside [0,%d]", b, n_tok[b], n_tok_max);
    if (max_frames[b] > N_CTX) return fail(WCA_ERR_TOO_LONG, "max_frames[%d]=%d > %d", b, max_frames[b], N_CTX);
    if (max_frames[b] < 1) return fail(WCA_ERR_INVALID, "max_frames[%d]=%d < 1", b, max_frames[b]);
    Fmax = max_frames[b] > Fmax ? max_frames[b] : Fmax;
  }
  *Fmax_out = Fmax;
  return WCA_OK;
}

// scores/top-k/aggregate/DTW on a dense weights tensor [B][LH][n_max][Fmax] whose column norms and
// scores are already in e->colnorm / e->scores.
struct Remat {
  const float* qk = nullptr;
  long qk_bs = 0, qk_hs = 0;
  int qk_ld = 0;
  const float* rowstats = nullptr;
};

int run_select_aggregate_dtw(wca_engine* e, const float* weights, int B, int LH, int n_max, int Fmax, const int* n_tok_dev,
                             const int* n_frames_dev, const int* dtwN_dev, const wca_align_opts* o, int L_layers,
                             const Remat* rm = nullptr, hipStream_t s_in = nullptr) {
  hipStream_t s = s_in ? s_in : e->stream;
  const int k = o->aggregation == WCA_AGGR_TOPK ? o->topk : 0;
  if (o->aggregation == WCA_AGGR_TOPK) {
    HIPCHK(e->sel.ensure(sizeof(int) * (size_t)B * k));
    HIPCHK(e->selsc.ensure(sizeof(float) * (size_t)B * k));
    HIPCHK(launch_topk((const float*)e->scores.p, LH, B, k, (int*)e->sel.p, (float*)e->selsc.p, s));
  }
  HIPCHK(e->matrix.ensure(sizeof(float) * (size_t)B * n_max * Fmax));
  AggregateArgs g{};
  g.weights = weights;
  g.w_bs = (long)LH * n_max * Fmax;
  g.n_tok_max = n_max;
  g.n_frames_max = Fmax;
  g.colnorm = (const float*)e->colnorm.p;
  g.LH = LH;
  g.B = B;
  g.n_tok = n_tok_dev;
  g.n_frames = n_frames_dev;
  g.row_lo = o->sot_len;
  g.row_hi_trim = 1;
  g.matrix = (float*)e->matrix.p;
  if (rm) {
    g.qk = rm->qk;
    g.qk_bs = rm->qk_bs;
    g.qk_hs = rm->qk_hs;
    g.qk_ld = rm->qk_ld;
    g.rowstats = rm->rowstats;
    g.medfilt_width = o->medfilt_width;
    g.qk_scale = o->qk_scale;
  }
  if (o->aggregation == WCA_AGGR_TOPK) {
    g.sel_idx = (const int*)e->sel.p;
    g.n_sel = k;
  } else {
    g.sel_idx = nullptr;
    const int H = LH / L_layers;
    g.head_lo = (L_layers / 2) * H;  // ws[n_layers//2:]  (timing.py:88)
  }
  HIPCHK(launch_aggregate(g, s));
  record(e, 6, s);

  const int Nmax = n_max - o->sot_len - 1;
  if (Nmax >= 1) {
    const int wpr = (Fmax + 15) / 16;
    const int cap = Nmax + Fmax + 2;
    HIPCHK(e->trace.ensure(sizeof(uint32_t) * (size_t)B * Nmax * wpr));
    HIPCHK(e->path.ensure(sizeof(int) * (size_t)B * 2 * cap));
    HIPCHK(e->pathlen.ensure(sizeof(int) * (size_t)B));
    HIPCHK(e->jump.ensure(sizeof(int) * (size_t)B * n_max));
    // the DTW writes n_tok[b] - sot_len - 1 entries per utterance; the rest of a row is defined as 0 (the buffer is recycled memory, and a
    // caller that compares or stores whole rows must not see what an earlier allocation left there)
    HIPCHK(hipMemsetAsync(e->jump.p, 0, sizeof(int) * (size_t)B * n_max, s));
    DtwArgs dg{};
    dg.matrix = (const float*)e->matrix.p;
    dg.m_bs = (long)n_max * Fmax;
    dg.ld = Fmax;
    dg.N = dtwN_dev;
    dg.M = n_frames_dev;
    dg.N_max = Nmax;
    dg.M_max = Fmax;
    dg.trace = (uint32_t*)e->trace.p;
    dg.path = (int*)e->path.p;
    dg.path_len = (int*)e->pathlen.p;
    dg.jump_frame = (int*)e->jump.p;
    dg.jump_ld = n_max;
    dg.P = B;
    HIPCHK(launch_dtw(dg, s));
  }
  return WCA_OK;
}

int ensure_res_host(wca_engine* e, int slot, size_t ints) {
  if (ints <= e->res_host_ints[slot]) return WCA_OK;
  if (e->res_host[slot]) (void)hipHostFree(e->res_host[slot]);
  e->res_host[slot] = nullptr;
  e->res_host_ints[slot] = 0;
  HIPCHK(hipHostMalloc((void**)&e->res_host[slot], ints * sizeof(int), hipHostMallocDefault));
  e->res_host_ints[slot] = ints;
  return WCA_OK;
}

}  // namespace

// =================================================================================== C ABI
extern "C" {

const char* wca_last_error(void) { return g_err.c_str(); }
int wca_version(void) { return 5; }   // (round 5: a new engine is in the contract mode; wca_engine_create_ex, W_lo slab, switch table)

int wca_engine_create(const wca_model_dims* dims, int device_ordinal, int max_batch, wca_engine** out) {
  return wca_engine_create_ex(dims, device_ordinal, max_batch, WCA_PRECISION_REFERENCE, out);   // the CONTRACT mode is the default (round 5)
}

int wca_engine_create_ex(const wca_model_dims* dims, int device_ordinal, int max_batch, int precision_mode, wca_engine** out) {
  if (!dims || !out) return fail(WCA_ERR_INVALID, "null argument");
  if (precision_mode != WCA_PRECISION_F16 && precision_mode != WCA_PRECISION_REFERENCE) return fail(WCA_ERR_INVALID, "precision mode %d", precision_mode);
  if (max_batch < 1 || max_batch > 256) return fail(WCA_ERR_INVALID, "max_batch %d outside [1,256]", max_batch);
  const wca_model_dims& D = *dims;
  if (D.n_audio_ctx != N_CTX || D.n_text_ctx != MAX_TOK) return fail(WCA_ERR_INVALID, "n_audio_ctx must be 1500 and n_text_ctx 448");
  if (D.n_audio_state % 128 || D.n_text_state % 128 || D.n_audio_state / D.n_audio_head != 64 || D.n_text_state / D.n_text_head != 64 ||
      D.n_audio_state > 1280 || D.n_text_state > 1280)
    return fail(WCA_ERR_INVALID, "unsupported widths: need n_state %% 128 == 0, n_state <= 1280, head_dim == 64");
  if (D.n_audio_state != D.n_text_state) return fail(WCA_ERR_INVALID, "n_audio_state != n_text_state");
  if (D.n_mels % 8 || D.n_mels > 256) return fail(WCA_ERR_INVALID, "n_mels must be a multiple of 8 and <= 256");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device_ordinal < 0 || device_ordinal >= ndev) return fail(WCA_ERR_INVALID, "device %d not present (%d devices)", device_ordinal, ndev);
  HIPCHK(hipSetDevice(device_ordinal));
  wca_engine* e = new wca_engine();
  e->dims = D;
  e->device = device_ordinal;
  e->max_batch = max_batch;
  HIPCHK(hipDeviceGetAttribute(&e->n_cu, hipDeviceAttributeMultiprocessorCount, device_ordinal));
  HIPCHK(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&e->stream3, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
  e->stream = e->own_stream;
  for (auto& ev : e->ev_kv) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  e->wslab_bytes = layout_weights(e, nullptr);
  HIPCHK(hipMalloc((void**)&e->wslab, e->wslab_bytes));
  HIPCHK(hipMemset(e->wslab, 0, e->wslab_bytes));
  layout_weights(e, e->wslab);
  if (precision_mode == WCA_PRECISION_REFERENCE) {   // born in the contract mode: the wide arena and the K-doubled weight copies from the start
    e->split = true;
    e->sites = (unsigned)WCA_PSITE_ALL;
    e->enc_from = 0;
    const size_t wbytes = layout_split_weights(e, nullptr);
    HIPCHK(hipMalloc((void**)&e->wslab2, wbytes));
    HIPCHK(hipMemset(e->wslab2, 0, wbytes));
    layout_split_weights(e, e->wslab2);
    e->sw_dirty = true;
  }
  const size_t abytes = layout_arena(e, nullptr);
  HIPCHK(hipMalloc((void**)&e->aslab, abytes));
  HIPCHK(hipMemset(e->aslab, 0, abytes));  // zero pad rows of mel_tm / h1pad and all slack
  layout_arena(e, e->aslab);
  HIPCHK(hipHostMalloc((void**)&e->meta_host, sizeof(int) * META_SLOTS * 4 * max_batch, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&e->err_host, sizeof(int) * 4, hipHostMallocDefault));
  e->err_host[0] = 0;
  for (auto& ev : e->ev) HIPCHK(hipEventCreate(&ev));
  for (auto& site : e->kev)
    for (auto& layer : site)
      for (auto& ev : layer) HIPCHK(hipEventCreate(&ev));
  for (auto& ev : e->res_ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  e->ev_valid = true;
  // constant tables of the STFT
  {
    std::vector<float> win(N_FFT), tw(2 * N_FFT);
    const double PI = 3.14159265358979323846;
    for (int n = 0; n < N_FFT; ++n) {
      win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * n / N_FFT));  // periodic hann
      tw[2 * n] = (float)std::cos(2.0 * PI * n / N_FFT);
      tw[2 * n + 1] = (float)std::sin(2.0 * PI * n / N_FFT);
    }
    HIPCHK(hipMemcpy(e->window, win.data(), sizeof(float) * N_FFT, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->twiddle, tw.data(), sizeof(float) * 2 * N_FFT, hipMemcpyHostToDevice));
  }
  *out = e;
  return WCA_OK;
}

void wca_engine_destroy(wca_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipDeviceSynchronize();
  if (e->part_s1) {
    e->enq_count = e->fetch_count;
    e->enc_q.clear();
    (void)wca_set_cu_partition(e, 0);
  }
  for (GrowBuf* g : {&e->cap, &e->wws, &e->colnorm, &e->scores, &e->sel, &e->selsc, &e->matrix, &e->trace, &e->path, &e->pathlen,
                     &e->jump, &e->tmp0, &e->tmp1})
    g->release();
  (void)wca_comm_destroy(e);
  e->coll_send.release();
  e->coll_recv.release();
  if (e->wslab) (void)hipFree(e->wslab);
  if (e->wslab2) (void)hipFree(e->wslab2);
  if (e->wslab_lo) (void)hipFree(e->wslab_lo);
  e->wlo_tmp[0].release();
  e->wlo_tmp[1].release();
  if (e->aslab) (void)hipFree(e->aslab);
  if (e->meta_host) (void)hipHostFree(e->meta_host);
  if (e->err_host) (void)hipHostFree(e->err_host);
  if (e->dec_done_host) (void)hipHostFree(e->dec_done_host);
  e->dec_cache.release();
  e->dec_tokens.release();
  e->dec_masks.release();
  e->dec_logits.release();
  e->dec_state.release();
  e->probe_jump.release();
  for (int i = 0; i < 2; ++i) {
    if (e->res_host[i]) (void)hipHostFree(e->res_host[i]);
    if (e->res_ev[i]) (void)hipEventDestroy(e->res_ev[i]);
  }
  if (e->ev_valid)
  {
    for (auto& ev : e->ev) (void)hipEventDestroy(ev);
    for (auto& site : e->kev)
      for (auto& layer : site)
        for (auto& ev : layer) (void)hipEventDestroy(ev);
  }
  for (auto& ev : e->ev_kv)
    if (ev) (void)hipEventDestroy(ev);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->stream3) (void)hipStreamDestroy(e->stream3);
  if (e->stream2) (void)hipStreamDestroy(e->stream2);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
}

int wca_engine_set_stream(wca_engine* e, void* hip_stream) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  e->user_stream = (hipStream_t)hip_stream;  // NULL is the HIP default (null) stream, which is what torch's default stream is
  e->user_stream_set = true;
  // CU-partitioned engine (wca_set_cu_partition): phase 1 stays on its masked stream, which is NOT ordered with the caller's stream
  // (WCA_STATUS_PARTITIONED tells the binding to synchronise around its calls); the stream recorded here is restored when the partition is lifted
  if (e->part_cus > 0) return WCA_STATUS_PARTITIONED;
  e->stream = e->user_stream;
  return WCA_OK;
}

int wca_engine_synchronize(wca_engine* e) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipStreamSynchronize(e->stream2));
  HIPCHK(hipStreamSynchronize(e->stream3));
  return WCA_OK;
}

int wca_set_decode_mode(wca_engine* e, int fused, int streams) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  if (streams != 1 && streams != 2) return fail(WCA_ERR_INVALID, "streams must be 1 or 2");
  e->dec_fused = fused != 0;
  e->dec_streams = streams;
  return WCA_OK;
}

int wca_set_profiling(wca_engine* e, int on) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  e->profiling = on != 0;
  return WCA_OK;
}

int wca_last_stage_ms(wca_engine* e, float* ms8) {
  if (!e || !ms8) return fail(WCA_ERR_INVALID, "null argument");
  if (!e->profiling) return fail(WCA_ERR_STATE, "profiling disabled");
  HIPCHK(hipEventSynchronize(e->ev[8]));
  for (int i = 0; i < 7; ++i) HIPCHK(hipEventElapsedTime(&ms8[i], e->ev[i], e->ev[i + 1]));
  HIPCHK(hipEventElapsedTime(&ms8[7], e->ev[0], e->ev[8]));
  return WCA_OK;
}

int wca_last_kernel_ms(wca_engine* e, int site, int* n_launches, float* total_ms, double* flops_per_launch, double* bytes_per_launch) {
  if (!e || !n_launches || !total_ms || !flops_per_launch || !bytes_per_launch) return fail(WCA_ERR_INVALID, "null argument");
  if (site < 0 || site >= WCA_N_SITES) return fail(WCA_ERR_INVALID, "site %d outside [0,%d)", site, WCA_N_SITES);
  if (!e->profiling) return fail(WCA_ERR_STATE, "profiling disabled");
  HIPCHK(hipEventSynchronize(e->ev[8]));
  int nl = 0;
  float tot = 0.f;
  for (int i = 0; i < 33; ++i) {
    if (!e->kev_set[site][i]) continue;
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->kev[site][i][0], e->kev[site][i][1]));
    tot += ms;
    ++nl;
  }
  *n_launches = nl;
  *total_ms = tot;
  // algorithmic work of ONE launch at the last batch size (M = batch * 1500 rows, d = n_audio_state). Bytes: operands read once, outputs
  // written once; an operand / output of a split site is an f16 PAIR (4 bytes per element instead of 2), the weights are read once
  const double d = e->dims.n_audio_state, M = (double)e->last_batch * N_CTX, H = e->dims.n_audio_head;
  const int lastl = e->dims.n_audio_layer - 1;
  const double pg = enc_gemm_split(e, lastl) ? 2.0 : 1.0, pa = enc_attn_split(e, lastl) ? 2.0 : 1.0;   // (the last block's flags: all blocks alike in the named modes)
  const double pin_out = (pg > 1.0 && pa > 1.0) ? 2.0 : 1.0;   // the out-projection multiplies pairs only behind a split attention
  double fl = 0, by = 0;
  switch (site) {
    case WCA_SITE_QKV: fl = 2 * M * 3 * d * d; by = 2 * (pg * M * d + 3 * d * d + pa * M * 3 * d); break;
    case WCA_SITE_ATTN: fl = 4.0 * e->last_batch * H * (double)N_CTX * N_CTX * 64; by = 2 * pa * (M * 3 * d + M * d); break;
    case WCA_SITE_OUT: fl = 2 * M * d * d; by = 2 * (pin_out * M * d + d * d) + 8 * M * d + (e->fuse_ln && pg == 1.0 ? 2 : 0) * M * d; break;  // f32 residual read + write (+ the fused LayerNorm's f16 output)
    case WCA_SITE_FC1: fl = 2 * M * 4 * d * d; by = 2 * (pg * M * d + 4 * d * d + pg * M * 4 * d); break;
    case WCA_SITE_FC2: fl = 2 * M * 4 * d * d; by = 2 * (pg * M * 4 * d + 4 * d * d) + 8 * M * d + (e->fuse_ln && pg == 1.0 ? 2 : 0) * M * d; break;
    case WCA_SITE_LN1:
    case WCA_SITE_LN2: fl = 8 * M * d; by = (4 + 2 * pg) * M * d; break;                         // read f32, write f16 (or the f16 pair)
  }
  *flops_per_launch = fl;
  *bytes_per_launch = by;
  return WCA_OK;
}

int wca_set_precision_sites(wca_engine* e, unsigned mask, int enc_first_layer) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  if (mask & ~(unsigned)WCA_PSITE_ALL) return fail(WCA_ERR_INVALID, "precision site mask 0x%x has unknown bits", mask);
  if (enc_first_layer < 0 || enc_first_layer > e->dims.n_audio_layer)
    return fail(WCA_ERR_INVALID, "enc_first_layer %d outside [0, %d]", enc_first_layer, e->dims.n_audio_layer);
  if (!(mask & (WCA_PSITE_ENC_GEMM | WCA_PSITE_ENC_ATTN))) enc_first_layer = 0;  // (unused: keep the state canonical)
  if (mask == e->sites && enc_first_layer == e->enc_from) return WCA_OK;
  if (e->enq_count != e->fetch_count) return fail(WCA_ERR_STATE, "fetch the batches in flight before changing the precision mode");
  for (auto& st : e->enc_q)
    if (!st.decoded) return fail(WCA_ERR_STATE, "an encoded batch is waiting: consume it before changing the precision mode");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipStreamSynchronize(e->stream2));
  HIPCHK(hipStreamSynchronize(e->stream3));
  const bool want = mask != 0;
  if (want != e->split) {
    // The activation arena is laid out per mode (operand buffers are twice as wide while any site is split) and the K-doubled
    // weight copies exist only then. Allocate the NEW arena (and copies) first; the old ones are released and the engine's
    // state committed only when both allocations succeeded, so a failed switch leaves a working engine in its previous mode.
    const bool was = e->split;
    e->split = want;
    const size_t abytes = layout_arena(e, nullptr);
    e->split = was;
    char* na = nullptr;
    char* nw = nullptr;
    size_t wbytes = 0;
    hipError_t he = hipMalloc((void**)&na, abytes);
    if (he == hipSuccess && want) {
      wbytes = layout_split_weights(e, nullptr);
      he = hipMalloc((void**)&nw, wbytes);
    }
    if (he == hipSuccess) he = hipMemset(na, 0, abytes);  // zero pad rows of mel_tm / h1pad, counters, flags and all slack
    if (he == hipSuccess && nw) he = hipMemset(nw, 0, wbytes);  // K padding of the conv1 copy stays zero
    if (he == hipSuccess && debug_switch(DBG_FAIL_PRECISION_ALLOC)) he = hipErrorOutOfMemory;  // fault injection (wca_test_set_switch) for the test of the path below
    if (he != hipSuccess) {
      if (na) (void)hipFree(na);
      if (nw) (void)hipFree(nw);
      (void)hipGetLastError();
      layout_arena(e, e->aslab);  // (the sizing pass above moved the arena pointers: restore them)
      if (was) layout_split_weights(e, e->wslab2);
      return fail(WCA_ERR_HIP, "precision switch: allocating the %s arena (%zu + %zu bytes) failed: %s; the engine keeps its previous mode",
                  want ? "wide" : "narrow", abytes, wbytes, hipGetErrorString(he));
    }
    (void)hipFree(e->aslab);
    if (e->wslab2) (void)hipFree(e->wslab2);
    e->aslab = na;
    e->wslab2 = nw;
    e->split = want;
    layout_arena(e, e->aslab);
    if (want) {
      layout_split_weights(e, e->wslab2);
      e->sw_dirty = true;  // built on the next entry point that runs the model (after the weights are final)
    }
    e->ln_err = nullptr;
  } else if (want) {
    // same arena, other row layouts inside it (a buffer's rows are [hi | lo] or single per site): the zero pad rows of the
    // time-major conv images move with the row width, so wipe the arena once
    const size_t abytes = layout_arena(e, nullptr);
    layout_arena(e, e->aslab);
    HIPCHK(hipMemset(e->aslab, 0, abytes));
  }
  for (auto& st : e->enc_q) e->slot_busy[st.slot] = false;  // decoded-but-never-aligned states die with the arena / its layout
  e->enc_q.clear();
  e->sites = mask;
  e->enc_from = enc_first_layer;
  return WCA_OK;
}

int wca_get_precision_sites(wca_engine* e, unsigned* mask_out, int* enc_first_layer_out) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  if (mask_out) *mask_out = e->sites;
  if (enc_first_layer_out) *enc_first_layer_out = e->enc_from;
  return WCA_OK;
}

int wca_set_precision(wca_engine* e, int mode) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  if (mode != WCA_PRECISION_F16 && mode != WCA_PRECISION_SPLIT) return fail(WCA_ERR_INVALID, "precision mode %d", mode);
  return wca_set_precision_sites(e, mode == WCA_PRECISION_SPLIT ? (unsigned)WCA_PSITE_ALL : 0u, 0);
}

int wca_get_precision(wca_engine* e) {
  if (!e || e->sites == 0) return WCA_PRECISION_F16;
  return (e->sites == (unsigned)WCA_PSITE_ALL && e->enc_from == 0) ? WCA_PRECISION_SPLIT : WCA_PRECISION_MIXED;
}

int wca_set_fuse_ln(wca_engine* e, int on) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  if (on && e->part_cus > 0)
    return fail(WCA_ERR_STATE, "the LayerNorm epilogue fusion needs one resident workgroup per CU of the whole device: not available while the CUs are partitioned (wca_set_cu_partition)");
  e->fuse_ln = on != 0;
  return WCA_OK;
}

int wca_set_cu_partition(wca_engine* e, int phase2_cus) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  if (phase2_cus < 0 || phase2_cus >= e->n_cu || (phase2_cus & 7)) return fail(WCA_ERR_INVALID, "phase2_cus %d: a multiple of 8 in [0, %d)", phase2_cus, e->n_cu);
  if (e->enq_count != e->fetch_count || !e->enc_q.empty()) return fail(WCA_ERR_STATE, "fetch / consume the batches in flight before changing the CU partition");
  if (phase2_cus > 0 && e->fuse_ln)
    return fail(WCA_ERR_STATE, "the LayerNorm epilogue fusion (wca_set_fuse_ln) needs every CU of the device: switch it off before partitioning the CUs");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipDeviceSynchronize());
  // the masked streams of the NEW partition are created first and committed only when all three exist (a failure half way
  // used to leave part_s1 set with nothing saved)
  hipStream_t ns[3] = {nullptr, nullptr, nullptr};
  if (phase2_cus > 0) {
    // mask bit i = CU i of the device's enumeration; the round-robin of mask bits over the 8 XCDs gives both partitions CUs on every XCD
    const int words = (e->n_cu + 31) / 32;
    std::vector<uint32_t> m1(words, 0u), m2(words, 0u);
    for (int i = 0; i < e->n_cu; ++i) (i < phase2_cus ? m2 : m1)[i >> 5] |= 1u << (i & 31);
    for (int i = 0; i < 3; ++i) {
      const hipError_t he = hipExtStreamCreateWithCUMask(&ns[i], (uint32_t)words, (i == 0 ? m1 : m2).data());
      if (he != hipSuccess) {
        for (int j = 0; j < i; ++j) (void)hipStreamDestroy(ns[j]);
        (void)hipGetLastError();
        return fail(WCA_ERR_HIP, "CU-masked stream %d: %s; the engine keeps its previous partition", i, hipGetErrorString(he));
      }
    }
  }
  if (e->part_s1) {   // lift the present partition: phase 1 goes back to the stream the CALLER bound (not to the engine's private one)
    e->stream = e->user_stream_set ? e->user_stream : e->own_stream;
    e->stream2 = e->saved_s2;
    e->stream3 = e->saved_s3;
    (void)hipStreamDestroy(e->part_s1);
    (void)hipStreamDestroy(e->part_s2);
    (void)hipStreamDestroy(e->part_s3);
    e->part_s1 = e->part_s2 = e->part_s3 = nullptr;
    e->saved_s2 = e->saved_s3 = nullptr;
  }
  e->part_cus = 0;
  if (phase2_cus == 0) return WCA_OK;
  e->part_s1 = ns[0];
  e->part_s2 = ns[1];
  e->part_s3 = ns[2];
  e->saved_s2 = e->stream2;
  e->saved_s3 = e->stream3;
  e->stream = e->part_s1;   // (wca_engine_set_stream only RECORDS the caller's stream while the partition is active: that stream has no CU mask)
  e->stream2 = e->part_s2;
  e->stream3 = e->part_s3;
  e->part_cus = phase2_cus;
  return WCA_OK;
}

int wca_set_overlap(wca_engine* e, int on) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  if (e->enq_count != e->fetch_count) return fail(WCA_ERR_STATE, "fetch the batches in flight before changing the stream layout");
  e->overlap = on != 0;
  return WCA_OK;
}

int wca_load_weight(wca_engine* e, const char* name_c, const void* p, int dtype, const int64_t* shape, int ndim) {
  if (!e || !name_c || !p || !shape) return fail(WCA_ERR_INVALID, "null argument");
  if (dtype != WCA_DTYPE_F32 && dtype != WCA_DTYPE_F16) return fail(WCA_ERR_INVALID, "dtype %d", dtype);
  HIPCHK(hipSetDevice(e->device));
  const std::string name(name_c);
  const size_t n = numel(shape, ndim);
  const wca_model_dims& D = e->dims;
  const int d = D.n_audio_state, dt = D.n_text_state;
  int rc = 1;
  size_t n_inexact = 0;   // elements of this tensor that its f16 storage rounded (put_f16 / put_conv)
#define WANTN(cnt) \
  if (n != (size_t)(cnt)) return fail(WCA_ERR_INVALID, "weight %s: expected %zu elements, got %zu", name_c, (size_t)(cnt), n)
  if (name == "mel_filters") {
    WANTN((size_t)D.n_mels * N_BIN);
    std::vector<float> f(n);
    for (size_t i = 0; i < n; ++i) f[i] = host_val(p, dtype, i);
    std::vector<int> lo(D.n_mels), hi(D.n_mels);
    for (int m = 0; m < D.n_mels; ++m) {
      int l = N_BIN, h = 0;
      for (int k = 0; k < N_BIN; ++k)
        if (f[(size_t)m * N_BIN + k] != 0.f) {
          l = k < l ? k : l;
          h = k + 1;
        }
      if (h == 0) l = 0;
      lo[m] = l;
      hi[m] = h;
    }
    HIPCHK(hipMemcpy(e->mel_filters, f.data(), n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->filt_lo, lo.data(), sizeof(int) * D.n_mels, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->filt_hi, hi.data(), sizeof(int) * D.n_mels, hipMemcpyHostToDevice));
    e->have_filters = true;
    rc = WCA_OK;
  } else if (name == "encoder.conv1.weight") {
    WANTN((size_t)d * D.n_mels * 3);
    rc = put_conv(e, e->conv1_w, p, dtype, d, D.n_mels, e->k1pad, &n_inexact);
  } else if (name == "encoder.conv1.bias") {
    WANTN(d);
    rc = put_f32(e->conv1_b, p, dtype, n);
  } else if (name == "encoder.conv2.weight") {
    WANTN((size_t)d * d * 3);
    rc = put_conv(e, e->conv2_w, p, dtype, d, d, 3 * d, &n_inexact);
  } else if (name == "encoder.conv2.bias") {
    WANTN(d);
    rc = put_f32(e->conv2_b, p, dtype, n);
  } else if (name == "encoder.positional_embedding") {
    WANTN((size_t)N_CTX * d);
    rc = put_f32(e->enc_pos, p, dtype, n);
  } else if (name == "encoder.ln_post.weight") {
    WANTN(d);
    rc = put_f32(e->lnpost_g, p, dtype, n);
  } else if (name == "encoder.ln_post.bias") {
    WANTN(d);
    rc = put_f32(e->lnpost_b, p, dtype, n);
  } else if (name == "decoder.token_embedding.weight") {
    WANTN((size_t)D.n_vocab * dt);
    rc = put_f16(e, e->tok_emb, e->tok_emb, p, dtype, n, &n_inexact);
  } else if (name == "decoder.positional_embedding") {
    WANTN((size_t)D.n_text_ctx * dt);
    rc = put_f32(e->dec_pos, p, dtype, n);
  } else if (name == "decoder.ln.weight") {
    WANTN(dt);
    rc = put_f32(e->lnf_g, p, dtype, n);
  } else if (name == "decoder.ln.bias") {
    WANTN(dt);
    rc = put_f32(e->lnf_b, p, dtype, n);
  } else if (name.rfind("encoder.blocks.", 0) == 0 || name.rfind("decoder.blocks.", 0) == 0) {
    const bool is_dec = name[0] == 'd';
    const size_t p0 = 15;
    const size_t dot = name.find('.', p0);
    if (dot == std::string::npos) return fail(WCA_ERR_INVALID, "bad weight name %s", name_c);
    const int li = atoi(name.substr(p0, dot - p0).c_str());
    const int nl = is_dec ? D.n_text_layer : D.n_audio_layer;
    if (li < 0 || li >= nl) return fail(WCA_ERR_INVALID, "layer index out of range in %s", name_c);
    rc = load_block_tensor(e, is_dec ? e->dec[li] : e->enc[li], is_dec, li, name.substr(dot + 1), p, dtype, n, is_dec ? dt : d, &n_inexact);
  }
#undef WANTN
  if (rc == WCA_OK) {
    e->loaded.insert(name);
    e->sw_dirty = true;
    if (n_inexact) e->inexact[name] = n_inexact;
    else e->inexact.erase(name);
    if (e->inexact.empty()) e->wlo_bases.clear();   // (every remainder in the W_lo slab is zero again)
  }
  return rc < 0 ? rc : WCA_OK;  // unknown names (e.g. alignment_heads) are ignored
}

int wca_finalize_weights(wca_engine* e) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  const wca_model_dims& D = e->dims;
  std::vector<std::string> need = {"encoder.conv1.weight", "encoder.conv1.bias", "encoder.conv2.weight", "encoder.conv2.bias",
                                   "encoder.positional_embedding", "encoder.ln_post.weight", "encoder.ln_post.bias",
                                   "decoder.token_embedding.weight", "decoder.positional_embedding", "decoder.ln.weight",
                                   "decoder.ln.bias"};
  const char* blk[] = {"attn.query.weight", "attn.query.bias", "attn.key.weight", "attn.value.weight", "attn.value.bias",
                       "attn.out.weight", "attn.out.bias", "attn_ln.weight", "attn_ln.bias", "mlp.0.weight", "mlp.0.bias",
                       "mlp.2.weight", "mlp.2.bias", "mlp_ln.weight", "mlp_ln.bias"};
  const char* cblk[] = {"cross_attn.query.weight", "cross_attn.query.bias", "cross_attn.key.weight", "cross_attn.value.weight",
                        "cross_attn.value.bias", "cross_attn.out.weight", "cross_attn.out.bias", "cross_attn_ln.weight",
                        "cross_attn_ln.bias"};
  for (int i = 0; i < D.n_audio_layer; ++i)
    for (const char* b : blk) need.push_back("encoder.blocks." + std::to_string(i) + "." + b);
  for (int i = 0; i < D.n_text_layer; ++i) {
    for (const char* b : blk) need.push_back("decoder.blocks." + std::to_string(i) + "." + b);
    for (const char* b : cblk) need.push_back("decoder.blocks." + std::to_string(i) + "." + b);
  }
  for (const auto& nm : need)
    if (!e->loaded.count(nm)) return fail(WCA_ERR_STATE, "missing weight %s", nm.c_str());
  e->finalized = true;
  return WCA_OK;
}

int wca_weights_inexact(wca_engine* e, long long* n_tensors_out, long long* n_values_out, char* first_name_out, int first_name_cap) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  long long tot = 0;
  for (const auto& kv : e->inexact) tot += (long long)kv.second;
  if (n_tensors_out) *n_tensors_out = (long long)e->inexact.size();
  if (n_values_out) *n_values_out = tot;
  if (first_name_out && first_name_cap > 0) {
    const std::string f = e->inexact.empty() ? std::string() : e->inexact.begin()->first;
    snprintf(first_name_out, (size_t)first_name_cap, "%s", f.c_str());
  }
  return WCA_OK;
}

int wca_set_allow_rounded_weights(wca_engine* e, int on) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  e->allow_rounded = on != 0;
  return WCA_OK;
}

int wca_log_mel(wca_engine* e, const float* pcm_dev, int64_t pcm_stride, const int32_t* n_samples_host, int batch, float* mel_out_dev) {
  if (!e || !pcm_dev || !n_samples_host || !mel_out_dev) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (int jr = join_phase2(e)) return jr;
  if (batch < 1 || batch > e->max_batch) return fail(WCA_ERR_INVALID, "batch %d outside [1,%d]", batch, e->max_batch);
  for (int b = 0; b < batch; ++b)
    if (n_samples_host[b] < 0 || n_samples_host[b] > 480000 || n_samples_host[b] > pcm_stride)
      return fail(WCA_ERR_INVALID, "n_samples[%d]=%d invalid (pad_or_trim to <= 480000 first)", b, n_samples_host[b]);
  int* rows[4];
  int rc = stage_meta(e, batch, n_samples_host, nullptr, nullptr, nullptr, rows);
  if (rc) return rc;
  return run_logmel(e, pcm_dev, pcm_stride, rows[0], batch, mel_out_dev, false);
}

int wca_get_attentions(wca_engine* e, const float* mel_dev, const int64_t* tokens_dev, int batch, int n_tok, const int32_t* n_tok_host,
                       const int32_t* max_frames_host, int medfilt_width, float qk_scale, float* weights_out_dev,
                       float* logits_out_dev) {
  int rc = check_ready(e);
  if (rc) return rc;
  if ((rc = join_phase2(e))) return rc;
  if (!mel_dev || !tokens_dev || !max_frames_host || !weights_out_dev) return fail(WCA_ERR_INVALID, "null argument");
  if (batch < 1 || batch > e->max_batch) return fail(WCA_ERR_INVALID, "batch %d outside [1,%d]", batch, e->max_batch);
  if (medfilt_width < 1 || !(medfilt_width & 1) || medfilt_width > 33) return fail(WCA_ERR_INVALID, "medfilt_width must be odd and <= 33");
  int Fmax = 0;
  rc = validate_lengths(batch, n_tok, n_tok_host, max_frames_host, &Fmax);
  if (rc) return rc;
  const wca_model_dims& D = e->dims;
  const int LH = D.n_text_layer * D.n_text_head;
  const int Fpad = (Fmax + 3) & ~3;
  std::vector<int32_t> ntok(batch);
  for (int b = 0; b < batch; ++b) ntok[b] = n_tok_host ? n_tok_host[b] : n_tok;
  int* rows[4];
  rc = stage_meta(e, batch, nullptr, ntok.data(), max_frames_host, nullptr, rows);
  if (rc) return rc;
  if ((rc = mel_to_tm(e, mel_dev, batch))) return rc;
  // cross-K/V go into a slot no queued batch (wca_encode_batch / wca_greedy_decode / an un-fetched alignment) still needs
  const int slot = take_kv_slot(e);
  if (slot < 0) return fail(WCA_ERR_STATE, "both cross-K/V slots hold live batches: fetch or consume one first");
  half_t* kvbuf = slot ? e->kv_alt : e->kv;
  HIPCHK(hipMemsetAsync(e->err_dev, 0, sizeof(int), e->stream));
  rc = run_encoder(e, batch);
  if (rc) return rc;
  rc = run_cross_kv(e, batch, kvbuf);
  if (rc) return rc;
  HIPCHK(e->cap.ensure(sizeof(float) * (size_t)batch * LH * n_tok * Fpad));
  rc = run_decoder(e, tokens_dev, batch, n_tok, (float*)e->cap.p, Fpad, Fmax, logits_out_dev, nullptr, kvbuf);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(e->err_host, e->err_dev, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e->colnorm.ensure(sizeof(float) * (size_t)batch * LH * Fmax));
  HIPCHK(e->scores.ensure(sizeof(float) * (size_t)batch * LH));
  HeadStatsArgs h{};
  h.qk = (const float*)e->cap.p;
  h.qk_bs = (long)LH * n_tok * Fpad;
  h.qk_hs = (long)n_tok * Fpad;
  h.qk_ld = Fpad;
  h.weights = weights_out_dev;
  h.w_bs = (long)LH * n_tok * Fmax;
  h.n_tok = rows[1];
  h.n_frames = rows[2];
  h.n_tok_max = n_tok;
  h.n_frames_max = Fmax;
  h.colnorm = (float*)e->colnorm.p;
  h.scores = (float*)e->scores.p;
  h.LH = LH;
  h.B = batch;
  h.medfilt_width = medfilt_width;
  h.qk_scale = qk_scale;
  h.w_col = 1.f;
  h.w_row = 1.f;
  h.w_cov = 0.f;
  HIPCHK(launch_head_stats(h, e->stream));
  // this entry point is the reference's synchronous per-utterance call: the host learns here whether a token id was
  // outside the vocabulary (the row was embedded as token 0, never read out of bounds)
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->err_host[0] & 2) return fail(WCA_ERR_HIP, "LayerNorm statistics hand-off timed out inside a GEMM epilogue (a workgroup of a row panel never arrived)");
  if (e->err_host[0]) return fail(WCA_ERR_INVALID, "a token id is outside the model's vocabulary [0, %d) (tokenizer / checkpoint mismatch?)", D.n_vocab);
  return WCA_OK;
}

int wca_median_filter(wca_engine* e, const float* in_dev, float* out_dev, int64_t rows, int F, int width) {
  if (!e || !in_dev || !out_dev) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (int jr = join_phase2(e)) return jr;
  if (width < 1 || !(width & 1) || width > 33) return fail(WCA_ERR_INVALID, "filter width must be odd and <= 33");
  HIPCHK(launch_median_filter(in_dev, out_dev, rows, F, width, e->stream));
  return WCA_OK;
}

static int stats_on_weights(wca_engine* e, const float* attns_dev, int L, int H, int n, int F, float wc, float wr, float wv, int** rows_out,
                            int dtwN) {
  const int LH = L * H;
  if (L < 1 || H < 1 || n < 1 || n > MAX_TOK) return fail(WCA_ERR_INVALID, "bad shape L=%d H=%d n=%d", L, H, n);
  if (F < 1 || F > N_CTX) return fail(WCA_ERR_TOO_LONG, "F=%d outside [1,%d]", F, N_CTX);
  int32_t nt = n, nf = F, dn = dtwN;
  int rc = stage_meta(e, 1, nullptr, &nt, &nf, &dn, rows_out);
  if (rc) return rc;
  HIPCHK(e->colnorm.ensure(sizeof(float) * (size_t)LH * F));
  HIPCHK(e->scores.ensure(sizeof(float) * (size_t)LH));
  HeadStatsArgs h{};
  h.qk = attns_dev;
  h.qk_bs = 0;
  h.qk_hs = (long)n * F;
  h.qk_ld = F;
  h.weights = nullptr;
  h.n_tok = rows_out[1];
  h.n_frames = rows_out[2];
  h.n_tok_max = n;
  h.n_frames_max = F;
  h.colnorm = (float*)e->colnorm.p;
  h.scores = (float*)e->scores.p;
  h.LH = LH;
  h.B = 1;
  h.medfilt_width = 1;
  h.qk_scale = 1.f;
  h.w_col = wc;
  h.w_row = wr;
  h.w_cov = wv;
  h.input_is_weights = 1;
  HIPCHK(launch_head_stats(h, e->stream));
  return WCA_OK;
}

int wca_filter_attention(wca_engine* e, const float* attns_dev, int L, int H, int n, int F, int topk, float w_colnorm, float w_rownorm,
                         float w_coverage, float* scores_host, int32_t* sel_idx_host, float* sel_score_host) {
  if (!e || !attns_dev) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (int jr = join_phase2(e)) return jr;
  if (topk < 1) return fail(WCA_ERR_INVALID, "topk must be > 0");
  int* rows[4];
  int rc = stats_on_weights(e, attns_dev, L, H, n, F, w_colnorm, w_rownorm, w_coverage, rows, 0);
  if (rc) return rc;
  const int LH = L * H;
  const int keff = topk < LH ? topk : LH;
  HIPCHK(e->sel.ensure(sizeof(int) * (size_t)topk));
  HIPCHK(e->selsc.ensure(sizeof(float) * (size_t)topk));
  HIPCHK(launch_topk((const float*)e->scores.p, LH, 1, topk, (int*)e->sel.p, (float*)e->selsc.p, e->stream));
  if (scores_host) HIPCHK(hipMemcpyAsync(scores_host, e->scores.p, sizeof(float) * LH, hipMemcpyDeviceToHost, e->stream));
  if (sel_idx_host) HIPCHK(hipMemcpyAsync(sel_idx_host, e->sel.p, sizeof(int) * keff, hipMemcpyDeviceToHost, e->stream));
  if (sel_score_host) HIPCHK(hipMemcpyAsync(sel_score_host, e->selsc.p, sizeof(float) * keff, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return WCA_OK;
}

int wca_force_align(wca_engine* e, const float* ws_dev, int L, int H, int n, int F, const wca_align_opts* o, float* matrix_host,
                    int32_t* text_idx_host, int32_t* time_idx_host, int32_t* path_len_host, int32_t* sel_idx_host,
                    float* sel_score_host) {
  if (!e || !ws_dev || !o || !path_len_host) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (int jr = join_phase2(e)) return jr;
  if (o->aggregation != WCA_AGGR_MEAN && o->aggregation != WCA_AGGR_TOPK) return fail(WCA_ERR_INVALID, "aggregation %d", o->aggregation);
  if (o->aggregation == WCA_AGGR_TOPK && o->topk < 1) return fail(WCA_ERR_INVALID, "topk must be > 0 (timing.py:92)");
  const int N = n - o->sot_len - 1;
  if (o->sot_len < 0 || N < 1) return fail(WCA_ERR_INVALID, "n=%d leaves no rows after the [sot_len:-1] slice", n);
  int* rows[4];
  int rc = stats_on_weights(e, ws_dev, L, H, n, F, o->w_colnorm, o->w_rownorm, o->w_coverage, rows, N);
  if (rc) return rc;
  rc = run_select_aggregate_dtw(e, ws_dev, 1, L * H, n, F, rows[1], rows[2], rows[3], o, L);
  if (rc) return rc;
  const int cap = N + F + 2;
  std::vector<int> path(2 * (size_t)cap);
  int plen = 0;
  HIPCHK(hipMemcpyAsync(&plen, e->pathlen.p, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(path.data(), e->path.p, sizeof(int) * 2 * cap, hipMemcpyDeviceToHost, e->stream));
  if (matrix_host) HIPCHK(hipMemcpyAsync(matrix_host, e->matrix.p, sizeof(float) * (size_t)N * F, hipMemcpyDeviceToHost, e->stream));
  if (o->aggregation == WCA_AGGR_TOPK) {
    const int keff = o->topk < L * H ? o->topk : L * H;
    if (sel_idx_host) HIPCHK(hipMemcpyAsync(sel_idx_host, e->sel.p, sizeof(int) * keff, hipMemcpyDeviceToHost, e->stream));
    if (sel_score_host) HIPCHK(hipMemcpyAsync(sel_score_host, e->selsc.p, sizeof(float) * keff, hipMemcpyDeviceToHost, e->stream));
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  *path_len_host = plen;
  if (text_idx_host && time_idx_host)
    for (int i = 0; i < plen; ++i) {
      text_idx_host[i] = path[cap - plen + i];
      time_idx_host[i] = path[cap + cap - plen + i];
    }
  return WCA_OK;
}

static int dtw_dev_common(wca_engine* e, const float* matrix_dev, int P, int N, int M, bool want_jump) {
  if (N < 1 || N > 512 || M < 1 || M > 4096) return fail(WCA_ERR_INVALID, "DTW shape N=%d M=%d unsupported (N<=512, M<=4096)", N, M);
  const int wpr = (M + 15) / 16, cap = N + M + 2;
  HIPCHK(e->trace.ensure(sizeof(uint32_t) * (size_t)P * N * wpr));
  HIPCHK(e->path.ensure(sizeof(int) * (size_t)P * 2 * cap));
  HIPCHK(e->pathlen.ensure(sizeof(int) * (size_t)P));
  if (want_jump) HIPCHK(e->jump.ensure(sizeof(int) * (size_t)P * N));
  DtwArgs dg{};
  dg.matrix = matrix_dev;
  dg.m_bs = (long)N * M;
  dg.ld = M;
  dg.N_all = N;
  dg.M_all = M;
  dg.N_max = N;
  dg.M_max = M;
  dg.trace = (uint32_t*)e->trace.p;
  dg.path = (int*)e->path.p;
  dg.path_len = (int*)e->pathlen.p;
  dg.jump_frame = want_jump ? (int*)e->jump.p : nullptr;
  dg.jump_ld = N;
  dg.P = P;
  HIPCHK(launch_dtw(dg, e->stream));
  return WCA_OK;
}

int wca_dtw(wca_engine* e, const float* matrix_host, int N, int M, int32_t* text_idx_host, int32_t* time_idx_host, int32_t* path_len_host) {
  if (!e || !matrix_host || !text_idx_host || !time_idx_host || !path_len_host) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (int jr = join_phase2(e)) return jr;
  if (N < 1 || M < 1) return fail(WCA_ERR_INVALID, "empty DTW matrix");
  HIPCHK(e->tmp0.ensure(sizeof(float) * (size_t)N * M));
  HIPCHK(hipMemcpyAsync(e->tmp0.p, matrix_host, sizeof(float) * (size_t)N * M, hipMemcpyHostToDevice, e->stream));
  int rc = dtw_dev_common(e, (const float*)e->tmp0.p, 1, N, M, false);
  if (rc) return rc;
  const int cap = N + M + 2;
  std::vector<int> path(2 * (size_t)cap);
  int plen = 0;
  HIPCHK(hipMemcpyAsync(&plen, e->pathlen.p, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(path.data(), e->path.p, sizeof(int) * 2 * cap, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  *path_len_host = plen;
  for (int i = 0; i < plen; ++i) {
    text_idx_host[i] = path[cap - plen + i];
    time_idx_host[i] = path[cap + cap - plen + i];
  }
  return WCA_OK;
}

int wca_dtw_batch_dev(wca_engine* e, const float* matrix_dev, int P, int N, int M, int32_t* jump_frame_host) {
  if (!e || !matrix_dev || !jump_frame_host) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (int jr = join_phase2(e)) return jr;
  if (P < 1) return fail(WCA_ERR_INVALID, "P < 1");
  int rc = dtw_dev_common(e, matrix_dev, P, N, M, true);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(jump_frame_host, e->jump.p, sizeof(int) * (size_t)P * N, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return WCA_OK;
}

int wca_probe_heads(wca_engine* e, const float* ws_dev, int L, int H, int n, int F, int sot_len, float* scores_host,
                    int32_t* jump_frame_host) {
  if (!e || !ws_dev || !jump_frame_host) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (int jr = join_phase2(e)) return jr;
  const int LH = L * H, N = n - sot_len - 1;
  if (sot_len < 0 || N < 1) return fail(WCA_ERR_INVALID, "n=%d leaves no rows after the [sot_len:-1] slice", n);
  int* rows[4];
  int rc = stats_on_weights(e, ws_dev, L, H, n, F, 1.f, 1.f, 0.f, rows, N);
  if (rc) return rc;
  // every head becomes its own "utterance": matrix_h = ws_h / ||ws_h||_col  (timing.py:84-89 with L = H = 1)
  HIPCHK(e->tmp1.ensure(sizeof(int) * 2 * (size_t)LH));
  std::vector<int> meta(2 * (size_t)LH);
  for (int i = 0; i < LH; ++i) {
    meta[i] = n;
    meta[LH + i] = F;
  }
  HIPCHK(hipMemcpyAsync(e->tmp1.p, meta.data(), sizeof(int) * 2 * LH, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));  // meta is a stack-lifetime host buffer
  HIPCHK(e->matrix.ensure(sizeof(float) * (size_t)LH * n * F));
  AggregateArgs g{};
  g.weights = ws_dev;
  g.w_bs = (long)n * F;
  g.n_tok_max = n;
  g.n_frames_max = F;
  g.colnorm = (const float*)e->colnorm.p;
  g.sel_idx = nullptr;
  g.head_lo = 0;
  g.LH = 1;
  g.B = LH;
  g.n_tok = (const int*)e->tmp1.p;
  g.n_frames = (const int*)e->tmp1.p + LH;
  g.row_lo = sot_len;
  g.row_hi_trim = 1;
  g.matrix = (float*)e->matrix.p;
  HIPCHK(launch_aggregate(g, e->stream));
  const int wpr = (F + 15) / 16, cap = N + F + 2;
  HIPCHK(e->trace.ensure(sizeof(uint32_t) * (size_t)LH * N * wpr));
  HIPCHK(e->path.ensure(sizeof(int) * (size_t)LH * 2 * cap));
  HIPCHK(e->pathlen.ensure(sizeof(int) * (size_t)LH));
  HIPCHK(e->jump.ensure(sizeof(int) * (size_t)LH * N));
  DtwArgs dg{};
  dg.matrix = (const float*)e->matrix.p;
  dg.m_bs = (long)n * F;
  dg.ld = F;
  dg.N_all = N;
  dg.M_all = F;
  dg.N_max = N;
  dg.M_max = F;
  dg.trace = (uint32_t*)e->trace.p;
  dg.path = (int*)e->path.p;
  dg.path_len = (int*)e->pathlen.p;
  dg.jump_frame = (int*)e->jump.p;
  dg.jump_ld = N;
  dg.P = LH;
  HIPCHK(launch_dtw(dg, e->stream));
  HIPCHK(hipMemcpyAsync(jump_frame_host, e->jump.p, sizeof(int) * (size_t)LH * N, hipMemcpyDeviceToHost, e->stream));
  if (scores_host) HIPCHK(hipMemcpyAsync(scores_host, e->scores.p, sizeof(float) * LH, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e->probe_jump.ensure(sizeof(int) * (size_t)LH * N));
  HIPCHK(hipMemcpyAsync(e->probe_jump.p, e->jump.p, sizeof(int) * (size_t)LH * N, hipMemcpyDeviceToDevice, e->stream));
  e->probe_LH = LH;
  e->probe_N = N;
  HIPCHK(hipStreamSynchronize(e->stream));
  return WCA_OK;
}

int wca_probe_strict_tp(wca_engine* e, int n_heads, const int32_t* word_end_row_host, int n_hyp, const double* ref_times_host, int n_ref,
                        const uint8_t* same_word_host, double tolerance, int32_t* tp_host) {
  if (!e || !tp_host || (n_hyp > 0 && !word_end_row_host) || (n_ref > 0 && !ref_times_host) || (n_hyp > 0 && n_ref > 0 && !same_word_host))
    return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (e->probe_LH <= 0) return fail(WCA_ERR_STATE, "wca_probe_strict_tp needs a preceding wca_probe_heads");
  if (n_heads != e->probe_LH) return fail(WCA_ERR_INVALID, "n_heads %d != the %d heads of the preceding wca_probe_heads", n_heads, e->probe_LH);
  if (n_hyp < 0 || n_ref < 0 || n_ref > 512) return fail(WCA_ERR_INVALID, "n_hyp=%d n_ref=%d outside [0, 512]", n_hyp, n_ref);
  for (int i = 0; i < n_hyp; ++i)
    if (word_end_row_host[i] < 0 || word_end_row_host[i] >= e->probe_N)
      return fail(WCA_ERR_INVALID, "word end row %d = %d outside the %d aligned token rows", i, word_end_row_host[i], e->probe_N);
  const int LH = e->probe_LH;
  const size_t b_wb = align_up(sizeof(int) * (size_t)std::max(n_hyp, 1), 256), b_y = align_up(sizeof(double) * (size_t)std::max(n_ref, 1), 256),
               b_eq = align_up((size_t)std::max(n_hyp * n_ref, 1), 256), b_tp = sizeof(int) * (size_t)LH;
  HIPCHK(e->tmp0.ensure(b_wb + b_y + b_eq + b_tp));
  char* base = (char*)e->tmp0.p;
  if (n_hyp) HIPCHK(hipMemcpyAsync(base, word_end_row_host, sizeof(int) * (size_t)n_hyp, hipMemcpyHostToDevice, e->stream));
  if (n_ref) HIPCHK(hipMemcpyAsync(base + b_wb, ref_times_host, sizeof(double) * (size_t)n_ref, hipMemcpyHostToDevice, e->stream));
  if (n_hyp && n_ref) HIPCHK(hipMemcpyAsync(base + b_wb + b_y, same_word_host, (size_t)n_hyp * n_ref, hipMemcpyHostToDevice, e->stream));
  HIPCHK(launch_probe_strict((const int*)e->probe_jump.p, e->probe_N, LH, (const int*)base, n_hyp, (const double*)(base + b_wb), n_ref,
                             (const unsigned char*)(base + b_wb + b_y), tolerance, (int*)(base + b_wb + b_y + b_eq), e->stream));
  HIPCHK(hipMemcpyAsync(tp_host, base + b_wb + b_y + b_eq, b_tp, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return WCA_OK;
}

int wca_attention_weights(wca_engine* e, const float* qk_dev, int L, int H, int n, int ld, int max_frames, int medfilt_width,
                          float qk_scale, float* weights_out_dev) {
  if (!e || !qk_dev || !weights_out_dev) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (int jr = join_phase2(e)) return jr;
  const int LH = L * H;
  if (L < 1 || H < 1 || n < 1) return fail(WCA_ERR_INVALID, "bad shape L=%d H=%d n=%d", L, H, n);
  if (n > MAX_TOK) return fail(WCA_ERR_TOO_LONG, "n=%d > %d", n, MAX_TOK);
  if (max_frames < 1 || ld < max_frames) return fail(WCA_ERR_INVALID, "max_frames=%d must be in [1, ld=%d]", max_frames, ld);
  if (max_frames > N_CTX) return fail(WCA_ERR_TOO_LONG, "max_frames=%d > %d", max_frames, N_CTX);
  if (medfilt_width < 1 || !(medfilt_width & 1) || medfilt_width > 33) return fail(WCA_ERR_INVALID, "medfilt_width must be odd and <= 33");
  int32_t nt = n, nf = max_frames;
  int* rows[4];
  int rc = stage_meta(e, 1, nullptr, &nt, &nf, nullptr, rows);
  if (rc) return rc;
  HIPCHK(e->colnorm.ensure(sizeof(float) * (size_t)LH * max_frames));
  HIPCHK(e->scores.ensure(sizeof(float) * (size_t)LH));
  HeadStatsArgs h{};
  h.qk = qk_dev;
  h.qk_bs = 0;
  h.qk_hs = (long)n * ld;
  h.qk_ld = ld;
  h.weights = weights_out_dev;
  h.w_bs = 0;
  h.n_tok = rows[1];
  h.n_frames = rows[2];
  h.n_tok_max = n;
  h.n_frames_max = max_frames;
  h.colnorm = (float*)e->colnorm.p;
  h.scores = (float*)e->scores.p;
  h.LH = LH;
  h.B = 1;
  h.medfilt_width = medfilt_width;
  h.qk_scale = qk_scale;
  h.w_col = 1.f;
  h.w_row = 1.f;
  h.w_cov = 0.f;
  HIPCHK(launch_head_stats(h, e->stream));
  return WCA_OK;
}

int wca_default_find_alignment(wca_engine* e, const float* ws_dev, int L, int H, int n, int F, const int32_t* heads_host, int n_heads,
                                int sot_len, float* weights_norm_out_dev, float* matrix_host, int32_t* text_idx_host,
                                int32_t* time_idx_host, int32_t* path_len_host) {
  if (!e || !ws_dev || !heads_host || !path_len_host) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (int jr = join_phase2(e)) return jr;
  const int LH = L * H, N = n - sot_len - 1;
  if (n_heads < 1) return fail(WCA_ERR_INVALID, "empty alignment head list");
  if (L < 1 || H < 1 || n < 1 || n > MAX_TOK) return fail(WCA_ERR_INVALID, "bad shape L=%d H=%d n=%d", L, H, n);
  if (F < 1 || F > N_CTX) return fail(WCA_ERR_TOO_LONG, "F=%d outside [1,%d]", F, N_CTX);
  if (sot_len < 0 || N < 1) return fail(WCA_ERR_INVALID, "n=%d leaves no rows after the [sot_len:-1] slice", n);
  for (int i = 0; i < n_heads; ++i)
    if (heads_host[i] < 0 || heads_host[i] >= LH) return fail(WCA_ERR_INVALID, "alignment head %d out of range", heads_host[i]);
  // (w - mean) / std per head and frame over the token axis (two passes, population std), kept for the caller when it
  // asks for it (the reference returns these normalised weights, timing.py:186), then the mean over the heads
  const size_t norm_elems = (size_t)n_heads * n * F;
  HIPCHK(e->tmp1.ensure(sizeof(int) * (size_t)n_heads + (weights_norm_out_dev ? 0 : sizeof(float) * norm_elems) + 256));
  int* sel_dev = reinterpret_cast<int*>(e->tmp1.p);
  float* norm = weights_norm_out_dev ? weights_norm_out_dev
                                     : reinterpret_cast<float*>(reinterpret_cast<char*>(e->tmp1.p) + align_up(sizeof(int) * (size_t)n_heads, 256));
  HIPCHK(hipMemcpyAsync(sel_dev, heads_host, sizeof(int) * n_heads, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));  // heads_host is caller-owned pageable memory
  HIPCHK(launch_stdmean_normalize(ws_dev, sel_dev, n_heads, n, F, norm, e->stream));
  HIPCHK(e->matrix.ensure(sizeof(float) * (size_t)n * F));
  HIPCHK(launch_mean_heads(norm, n_heads, n, F, sot_len, 1, (float*)e->matrix.p, e->stream));
  int rc = dtw_dev_common(e, (const float*)e->matrix.p, 1, N, F, false);
  if (rc) return rc;
  const int cap = N + F + 2;
  std::vector<int> path(2 * (size_t)cap);
  int plen = 0;
  HIPCHK(hipMemcpyAsync(&plen, e->pathlen.p, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(path.data(), e->path.p, sizeof(int) * 2 * cap, hipMemcpyDeviceToHost, e->stream));
  if (matrix_host) HIPCHK(hipMemcpyAsync(matrix_host, e->matrix.p, sizeof(float) * (size_t)N * F, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  *path_len_host = plen;
  if (text_idx_host && time_idx_host)
    for (int i = 0; i < plen; ++i) {
      text_idx_host[i] = path[cap - plen + i];
      time_idx_host[i] = path[cap + cap - plen + i];
    }
  return WCA_OK;
}

int wca_align_batch_enqueue(wca_engine* e, const float* pcm_dev, int64_t pcm_stride, const int32_t* n_samples_host,
                            const int64_t* tokens_dev, int n_tok_max, const int32_t* n_tok_host, const int32_t* max_frames_host,
                            int batch, const wca_align_opts* o) {
  int rc = check_ready(e);
  if (rc) return rc;
  if (!tokens_dev || !n_tok_host || !max_frames_host || !o) return fail(WCA_ERR_INVALID, "null argument");
  const bool reuse_enc = (pcm_dev == nullptr);  // consume the oldest encoded state (wca_encode_batch / wca_greedy_decode)
  if (reuse_enc && (e->enc_q.empty() || e->enc_q.front().batch != batch))
    return fail(WCA_ERR_STATE, "pcm_dev == NULL re-uses the oldest state left by wca_encode_batch / wca_greedy_decode for the same batch; there is none");
  if (!reuse_enc && !n_samples_host) return fail(WCA_ERR_INVALID, "null argument");
  if (!reuse_enc && !e->enc_q.empty()) {
    // a stand-alone decode may have left a decoded state behind; an undecoded one is still wanted by its owner
    for (auto& st : e->enc_q)
      if (!st.decoded) return fail(WCA_ERR_STATE, "an encoded batch is waiting for wca_greedy_decode / wca_align_batch_enqueue(pcm_dev = NULL)");
    for (auto& st : e->enc_q) e->slot_busy[st.slot] = false;
    e->enc_q.clear();
  }
  if (e->enq_count - e->fetch_count >= 2) return fail(WCA_ERR_STATE, "two batches already in flight: call wca_align_batch_fetch first");
  if (batch < 1 || batch > e->max_batch) return fail(WCA_ERR_INVALID, "batch %d outside [1,%d]", batch, e->max_batch);
  if (o->aggregation != WCA_AGGR_MEAN && o->aggregation != WCA_AGGR_TOPK) return fail(WCA_ERR_INVALID, "aggregation %d", o->aggregation);
  if (o->aggregation == WCA_AGGR_TOPK && o->topk < 1) return fail(WCA_ERR_INVALID, "topk must be > 0 (timing.py:92)");
  if (o->medfilt_width < 1 || !(o->medfilt_width & 1) || o->medfilt_width > 33) return fail(WCA_ERR_INVALID, "medfilt_width must be odd and <= 33");
  int Fmax = 0;
  rc = validate_lengths(batch, n_tok_max, n_tok_host, max_frames_host, &Fmax);
  if (rc) return rc;
  if (!reuse_enc && (rc = check_pcm_lengths(n_samples_host, batch, pcm_stride))) return rc;
  const wca_model_dims& D = e->dims;
  const int LH = D.n_text_layer * D.n_text_head;
  const int Fpad = (Fmax + 3) & ~3;
  std::vector<int32_t> dn(batch);
  for (int b = 0; b < batch; ++b) {
    dn[b] = n_tok_host[b] - o->sot_len - 1;
    if (dn[b] < 0) dn[b] = 0;
  }
  int* rows[4];
  // (re-use: the metadata is only read by phase 2, so it travels on that stream -- `stream` may already hold the next
  // batch's phase 1, and an event recorded behind it would serialise this batch's phase 2 after it)
  hipStream_t s2 = e->overlap ? e->stream2 : e->stream;  // the stream phase 2 runs on
  rc = stage_meta(e, batch, reuse_enc ? nullptr : n_samples_host, n_tok_host, max_frames_host, dn.data(), rows,
                  reuse_enc ? s2 : nullptr);
  if (rc) return rc;
  // ---- phase 1 on `stream`: log-mel, encoder, cross-K/V of all decoder layers into a free K/V slot (a slot is busy
  // from its encode until the alignment that read it has been fetched; at most 2 alignments are in flight), or the
  // slot of the encoded state this call consumes.
  int bs;
  if (reuse_enc) {
    bs = e->enc_q.front().slot;
    e->enc_q.pop_front();
    record(e, 0);
    record(e, 1);
    record(e, 2);
    record(e, 3);
  } else {
    bs = take_kv_slot(e);
    if (bs < 0) return fail(WCA_ERR_STATE, "both cross-K/V slots hold live batches: fetch or consume one first");
    e->slot_busy[bs] = true;
    rc = run_phase1(e, nullptr, pcm_dev, pcm_stride, rows[0], batch, bs, /*skip_last_v=*/true);  // this path never asks for logits
    if (rc) {
      e->slot_busy[bs] = false;
      return rc;
    }
  }
  half_t* kvbuf = bs ? e->kv_alt : e->kv;
  // ---- phase 2 on `stream2`: decoder with capture, head statistics, top-k, aggregation, DTW, D2H. These are
  // latency-bound kernels with few workgroups; on their own stream they overlap the NEXT batch's phase 1.
  HIPCHK(hipStreamWaitEvent(s2, e->ev_kv[bs], 0));
  HIPCHK(hipMemsetAsync(e->err_dev, 0, sizeof(int), s2));
  HIPCHK(e->cap.ensure(sizeof(float) * (size_t)batch * LH * n_tok_max * Fpad));
  rc = run_decoder(e, tokens_dev, batch, n_tok_max, (float*)e->cap.p, Fpad, Fmax, nullptr, s2, kvbuf);
  if (rc) return rc;
  record(e, 4, s2);
  // the softmaxed maps are NOT materialised on this path (53 MB per utterance): head_stats keeps per-row
  // (max, sum) and the aggregation re-derives the values of the few selected heads from the captured logits
  HIPCHK(e->wws.ensure(sizeof(float) * (size_t)batch * LH * n_tok_max * 2));
  HIPCHK(e->colnorm.ensure(sizeof(float) * (size_t)batch * LH * Fmax));
  HIPCHK(e->scores.ensure(sizeof(float) * (size_t)batch * LH));
  HeadStatsArgs h{};
  h.qk = (const float*)e->cap.p;
  h.qk_bs = (long)LH * n_tok_max * Fpad;
  h.qk_hs = (long)n_tok_max * Fpad;
  h.qk_ld = Fpad;
  h.weights = nullptr;
  h.rowstats = (float*)e->wws.p;
  h.n_tok = rows[1];
  h.n_frames = rows[2];
  h.n_tok_max = n_tok_max;
  h.n_frames_max = Fmax;
  h.colnorm = (float*)e->colnorm.p;
  h.scores = (float*)e->scores.p;
  h.LH = LH;
  h.B = batch;
  h.medfilt_width = o->medfilt_width;
  h.qk_scale = o->qk_scale;
  h.w_col = o->w_colnorm;
  h.w_row = o->w_rownorm;
  h.w_cov = o->w_coverage;
  HIPCHK(launch_head_stats(h, s2));
  record(e, 5, s2);
  Remat rm;
  rm.qk = h.qk;
  rm.qk_bs = h.qk_bs;
  rm.qk_hs = h.qk_hs;
  rm.qk_ld = h.qk_ld;
  rm.rowstats = h.rowstats;
  rc = run_select_aggregate_dtw(e, nullptr, batch, LH, n_tok_max, Fmax, rows[1], rows[2], rows[3], o, D.n_text_layer, &rm, s2);
  if (rc) return rc;
  record(e, 7, s2);
  // results -> pinned staging (ring of 2 so the host can post-process batch i while batch i+1 runs)
  const int k = o->aggregation == WCA_AGGR_TOPK ? o->topk : 0;
  const int rs = (int)(e->enq_count & 1);
  rc = ensure_res_host(e, rs, (size_t)batch * n_tok_max + (size_t)batch * (k > 0 ? k : 1) + 2);
  if (rc) return rc;
  // the flags of this batch travel with its results (last two ints of the staging slot): phase 2's word, and the word phase 1
  // raised for this batch's cross-K/V slot (complete: s2 waited for ev_kv[bs], recorded behind that encoder)
  HIPCHK(hipMemcpyAsync(e->res_host[rs] + (size_t)batch * n_tok_max + (size_t)batch * (k > 0 ? k : 1), e->err_dev, sizeof(int),
                        hipMemcpyDeviceToHost, s2));
  HIPCHK(hipMemcpyAsync(e->res_host[rs] + (size_t)batch * n_tok_max + (size_t)batch * (k > 0 ? k : 1) + 1, e->err_dev + 1 + bs, sizeof(int),
                        hipMemcpyDeviceToHost, s2));
  if (n_tok_max - o->sot_len - 1 >= 1)
    HIPCHK(hipMemcpyAsync(e->res_host[rs], e->jump.p, sizeof(int) * (size_t)batch * n_tok_max, hipMemcpyDeviceToHost, s2));
  if (k > 0)
    HIPCHK(hipMemcpyAsync(e->res_host[rs] + (size_t)batch * n_tok_max, e->sel.p, sizeof(int) * (size_t)batch * k, hipMemcpyDeviceToHost, s2));
  record(e, 8, s2);
  HIPCHK(hipEventRecord(e->res_ev[rs], s2));
  e->res_batch[rs] = batch;
  e->res_ntok[rs] = n_tok_max;
  e->res_topk[rs] = k;
  e->res_kvslot[rs] = bs;
  e->last_batch = batch;
  e->enq_count++;
  return WCA_OK;
}

int wca_encode_batch(wca_engine* e, const float* mel_dev, const float* pcm_dev, int64_t pcm_stride, const int32_t* n_samples_host, int batch) {
  int rc = check_ready(e);
  if (rc) return rc;
  if ((mel_dev == nullptr) == (pcm_dev == nullptr)) return fail(WCA_ERR_INVALID, "pass exactly one of mel_dev / pcm_dev");
  if (pcm_dev && !n_samples_host) return fail(WCA_ERR_INVALID, "null argument");
  if (batch < 1 || batch > e->max_batch) return fail(WCA_ERR_INVALID, "batch %d outside [1,%d]", batch, e->max_batch);
  if (pcm_dev && (rc = check_pcm_lengths(n_samples_host, batch, pcm_stride))) return rc;
  const int slot = take_kv_slot(e);
  if (slot < 0) return fail(WCA_ERR_STATE, "both cross-K/V slots hold live batches: fetch or consume one first");
  int* rows[4] = {nullptr, nullptr, nullptr, nullptr};
  if (pcm_dev) {
    rc = stage_meta(e, batch, n_samples_host, nullptr, nullptr, nullptr, rows);
    if (rc) return rc;
  }
  e->slot_busy[slot] = true;
  rc = run_phase1(e, mel_dev, pcm_dev, pcm_stride, rows[0], batch, slot);
  if (rc) {
    e->slot_busy[slot] = false;
    return rc;
  }
  e->enc_q.push_back({slot, batch, false});
  return WCA_OK;
}

int wca_greedy_decode(wca_engine* e, const float* mel_dev, const float* pcm_dev, int64_t pcm_stride, const int32_t* n_samples_host,
                      int batch, const int32_t* initial_tokens_host, int n_initial, const uint8_t* suppress_mask_host,
                      const uint8_t* blank_mask_host, const wca_decode_opts* o, int32_t* tokens_out_host, int32_t* n_tokens_host,
                      float* sum_logprob_host, float* no_speech_prob_host) {
  int rc = check_ready(e);
  if (rc) return rc;
  if (mel_dev != nullptr && pcm_dev != nullptr) return fail(WCA_ERR_INVALID, "pass at most one of mel_dev / pcm_dev");
  const bool have_input = (mel_dev != nullptr) || (pcm_dev != nullptr);
  if (!initial_tokens_host || !suppress_mask_host || !o || !tokens_out_host || !n_tokens_host) return fail(WCA_ERR_INVALID, "null argument");
  if (pcm_dev && !n_samples_host) return fail(WCA_ERR_INVALID, "null argument");
  if (batch < 1 || batch > e->max_batch) return fail(WCA_ERR_INVALID, "batch %d outside [1,%d]", batch, e->max_batch);
  const wca_model_dims& D = e->dims;
  if (n_initial < 1 || o->sample_len < 1 || n_initial + o->sample_len > D.n_text_ctx)
    return fail(WCA_ERR_TOO_LONG, "n_initial %d + sample_len %d exceeds n_text_ctx %d", n_initial, o->sample_len, D.n_text_ctx);
  if (o->eot < 0 || o->eot >= D.n_vocab || o->timestamp_begin < 0 || o->timestamp_begin > D.n_vocab)
    return fail(WCA_ERR_INVALID, "eot / timestamp_begin outside the vocabulary");
  for (int i = 0; i < n_initial; ++i)
    if (initial_tokens_host[i] < 0 || initial_tokens_host[i] >= D.n_vocab) return fail(WCA_ERR_INVALID, "initial token %d outside the vocabulary", i);
  const int V = D.n_vocab, dt = D.n_text_state, L = D.n_text_layer;
  const int T_max = n_initial + o->sample_len;
  // ---- phase 1 on `stream` (unless an encoded state is waiting: wca_encode_batch); the state stays queued for the
  // alignment (wca_align_batch_enqueue with pcm_dev = NULL)
  // a state that was decoded but never aligned is stale once another decode starts (stand-alone whisper.decode use)
  for (auto it = e->enc_q.begin(); it != e->enc_q.end();) {
    if (it->decoded) {
      e->slot_busy[it->slot] = false;
      it = e->enc_q.erase(it);
    } else {
      ++it;
    }
  }
  if (have_input) {
    rc = wca_encode_batch(e, mel_dev, pcm_dev, pcm_stride, n_samples_host, batch);
    if (rc) return rc;
  }
  wca_engine::EncState* st = nullptr;
  for (auto& q : e->enc_q)
    if (!q.decoded) {
      st = &q;
      break;
    }
  if (!st || st->batch != batch) return fail(WCA_ERR_STATE, "no encoded batch of %d utterances is waiting to be decoded", batch);
  const int bs = st->slot;
  half_t* kvbuf = bs ? e->kv_alt : e->kv;
  // ---- the autoregressive loop on `stream2` (it shares the decoder scratch with phase 2 of the alignment, which is
  // ordered before it on that stream; phase 1 of the NEXT batch may run beside it on `stream`)
  hipStream_t s2 = e->stream2;
  HIPCHK(hipStreamWaitEvent(s2, e->ev_kv[bs], 0));
  HIPCHK(e->dec_cache.ensure(sizeof(half_t) * (size_t)L * 2 * batch * T_max * dt));
  HIPCHK(e->dec_tokens.ensure(sizeof(int) * (size_t)batch * T_max));
  HIPCHK(e->dec_masks.ensure((size_t)2 * V));
  HIPCHK(e->dec_logits.ensure(sizeof(float) * (size_t)batch * V));
  HIPCHK(e->dec_state.ensure(sizeof(float) * 2 * batch + sizeof(int) * (size_t)T_max));
  const bool want_nsp = no_speech_prob_host != nullptr && o->no_speech >= 0 && o->no_speech < V;
  if (!e->dec_done_host) HIPCHK(hipHostMalloc((void**)&e->dec_done_host, sizeof(int) * 4, hipHostMallocDefault));
  std::vector<int32_t> init((size_t)batch * T_max, o->eot);
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < n_initial; ++i) init[(size_t)b * T_max + i] = initial_tokens_host[i];
  int* tokens_dev = (int*)e->dec_tokens.p;
  unsigned char* masks = (unsigned char*)e->dec_masks.p;
  float* sum_lp = (float*)e->dec_state.p;
  float* nsp = sum_lp + batch;
  int* n_done = (int*)((char*)e->dec_state.p + sizeof(float) * 2 * batch);
  HIPCHK(hipMemcpyAsync(tokens_dev, init.data(), sizeof(int) * init.size(), hipMemcpyHostToDevice, s2));
  HIPCHK(hipMemcpyAsync(masks, suppress_mask_host, V, hipMemcpyHostToDevice, s2));
  if (blank_mask_host) HIPCHK(hipMemcpyAsync(masks + V, blank_mask_host, V, hipMemcpyHostToDevice, s2));
  HIPCHK(hipMemsetAsync(e->dec_state.p, 0, sizeof(float) * 2 * batch + sizeof(int) * (size_t)T_max, s2));
  HIPCHK(hipStreamSynchronize(s2));  // `init` is pageable host memory
  DecodeSelectArgs sel{};
  sel.logits = (const float*)e->dec_logits.p;
  sel.ld = V;
  sel.n_vocab = V;
  sel.tokens = tokens_dev;
  sel.T_max = T_max;
  sel.n_initial = n_initial;
  sel.suppress_mask = masks;
  sel.blank_mask = blank_mask_host ? masks + V : nullptr;
  sel.eot = o->eot;
  sel.timestamp_begin = o->timestamp_begin;
  sel.apply_timestamp_rules = o->apply_timestamp_rules;
  sel.max_initial_timestamp_index = o->max_initial_timestamp_index;
  sel.sum_logprob = sum_lp;
  sel.n_done = n_done;
  // the prompt is fed one position at a time (it is 3 tokens: sot, language, task); sampling starts after its last token.
  // The batch is decoded as two half-batches on two streams: a step is ~200 dependent launches of 5-10 us plus one
  // HBM-bound cross-attention per layer, and the halves are independent, so one half's small kernels run under the other
  // half's cross-K/V stream. Rows never interact (per-row kernels, per-row cache planes); n_done is an atomic counter.
  const int n_half = (e->dec_streams == 2 && batch >= 16) ? 2 : 1;
  const int hb[3] = {0, n_half == 2 ? (batch / 2 + 7) / 8 * 8 : batch, batch};
  hipStream_t hs[2] = {s2, e->stream3};
  if (n_half == 2) {
    HIPCHK(hipEventRecord(e->ev_fork, s2));
    HIPCHK(hipStreamWaitEvent(e->stream3, e->ev_fork, 0));
  }
  int steps = 0;
  static const bool dbg_host = std::getenv("WCA_DEC_DEBUG") != nullptr;   // (read once)
  double host_us = 0.0;
  for (int t = 0; t < T_max - 1; ++t) {
    const bool sample = (t >= n_initial - 1);
    const bool sot_logits = (t == 0 && want_nsp);  // probs_at_sot of DecodingTask._main_loop (sot_index = 0: no prompt)
    const auto h0 = std::chrono::steady_clock::now();
    for (int phase = -1; phase <= L; ++phase)
      for (int h = 0; h < n_half; ++h) {
        rc = run_decode_step(e, hs[h], h, kvbuf, tokens_dev, hb[h], hb[h + 1] - hb[h], batch, t, T_max, sample || sot_logits, phase);
        if (rc) return rc;
      }
    for (int h = 0; h < n_half; ++h) {
      const int b0 = hb[h], nb = hb[h + 1] - hb[h];
      if (sot_logits) HIPCHK(launch_token_prob((const float*)e->dec_logits.p + (size_t)b0 * V, V, V, o->no_speech, nsp + b0, nb, hs[h]));
      if (!sample) continue;
      DecodeSelectArgs sh = sel;
      sh.logits = sel.logits + (size_t)b0 * V;
      sh.tokens = sel.tokens + (size_t)b0 * T_max;
      sh.sum_logprob = sel.sum_logprob + b0;
      sh.cur_len = t + 1;
      HIPCHK(launch_decode_select(sh, nb, hs[h]));
    }
    if (dbg_host) host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
    if (!sample) continue;
    ++steps;
    // whisper's loop ends when every row has produced EOT (checked every 4 steps here: a late stop only costs time,
    // finished rows keep emitting EOT) or after sample_len steps
    if ((steps & 3) == 0 || steps == o->sample_len) {
      if (n_half == 2) {
        HIPCHK(hipEventRecord(e->ev_join, e->stream3));
        HIPCHK(hipStreamWaitEvent(s2, e->ev_join, 0));
      }
      HIPCHK(hipMemcpyAsync(e->dec_done_host, n_done + t + 1, sizeof(int), hipMemcpyDeviceToHost, s2));
      HIPCHK(hipStreamSynchronize(s2));
      if (e->dec_done_host[0] >= batch) break;
    }
    if (steps >= o->sample_len) break;
  }
  if (n_half == 2) {
    HIPCHK(hipEventRecord(e->ev_join, e->stream3));
    HIPCHK(hipStreamWaitEvent(s2, e->ev_join, 0));
  }
  if (dbg_host) fprintf(stderr, "[wca] greedy decode: host enqueue time %.1f us per position (%d halves)\n", host_us / (steps + n_initial - 1), n_half);
  std::vector<int32_t> toks((size_t)batch * T_max);
  HIPCHK(hipMemcpyAsync(toks.data(), tokens_dev, sizeof(int) * toks.size(), hipMemcpyDeviceToHost, s2));
  std::vector<float> lp(2 * (size_t)batch);
  HIPCHK(hipMemcpyAsync(lp.data(), sum_lp, sizeof(float) * 2 * batch, hipMemcpyDeviceToHost, s2));
  HIPCHK(hipStreamSynchronize(s2));
  const int n_have = n_initial + steps;  // positions written so far
  for (int b = 0; b < batch; ++b) {
    int n = n_have;
    for (int i = n_initial; i < n_have; ++i)
      if (toks[(size_t)b * T_max + i] == o->eot) {
        n = i;
        break;
      }
    n_tokens_host[b] = n;  // tokens_out[b][n_initial : n] are the sampled tokens before the first EOT
    for (int i = 0; i < T_max; ++i) tokens_out_host[(size_t)b * T_max + i] = (i < n_have) ? toks[(size_t)b * T_max + i] : o->eot;
    if (sum_logprob_host) sum_logprob_host[b] = lp[b];
    if (want_nsp) no_speech_prob_host[b] = lp[batch + b];
  }
  st->decoded = true;
  e->last_batch = batch;
  return WCA_OK;
}

int wca_align_batch_fetch(wca_engine* e, int batch, int n_tok_max, int topk, int32_t* jump_frame_host, int32_t* sel_idx_host) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  HIPCHK(hipSetDevice(e->device));
  if (e->fetch_count >= e->enq_count) return fail(WCA_ERR_STATE, "nothing to fetch");
  const int rs = (int)(e->fetch_count & 1);  // oldest un-fetched batch
  if (batch != e->res_batch[rs] || n_tok_max != e->res_ntok[rs]) return fail(WCA_ERR_STATE, "fetch does not match the oldest pending enqueue");
  if (sel_idx_host && e->res_topk[rs] > 0 && topk != e->res_topk[rs]) return fail(WCA_ERR_STATE, "topk does not match the pending enqueue");
  HIPCHK(hipEventSynchronize(e->res_ev[rs]));
  if (jump_frame_host) memcpy(jump_frame_host, e->res_host[rs], sizeof(int) * (size_t)batch * n_tok_max);
  if (sel_idx_host && e->res_topk[rs] > 0)
    memcpy(sel_idx_host, e->res_host[rs] + (size_t)batch * n_tok_max, sizeof(int) * (size_t)batch * topk);
  if (e->res_kvslot[rs] >= 0) e->slot_busy[e->res_kvslot[rs]] = false;
  e->res_kvslot[rs] = -1;
  e->fetch_count++;
  const int kk = e->res_topk[rs];
  const size_t fo = (size_t)batch * n_tok_max + (size_t)batch * (kk > 0 ? kk : 1);
  const int flag = e->res_host[rs][fo] | (e->res_host[rs][fo + 1] & 2);
  if (flag & 2) return fail(WCA_ERR_HIP, "LayerNorm statistics hand-off timed out inside a GEMM epilogue (a workgroup of a row panel never arrived)");
  if (flag)
    return fail(WCA_ERR_INVALID, "a token id is outside the model's vocabulary [0, %d) (tokenizer / checkpoint mismatch?)", e->dims.n_vocab);
  return WCA_OK;
}

int wca_align_batch(wca_engine* e, const float* pcm_dev, int64_t pcm_stride, const int32_t* n_samples_host, const int64_t* tokens_dev,
                    int n_tok_max, const int32_t* n_tok_host, const int32_t* max_frames_host, int batch, const wca_align_opts* o,
                    int32_t* jump_frame_host, int32_t* sel_idx_host) {
  int rc = wca_align_batch_enqueue(e, pcm_dev, pcm_stride, n_samples_host, tokens_dev, n_tok_max, n_tok_host, max_frames_host, batch, o);
  if (rc) return rc;
  return wca_align_batch_fetch(e, batch, n_tok_max, o->aggregation == WCA_AGGR_TOPK ? o->topk : 0, jump_frame_host, sel_idx_host);
}

// ---------------------------------------------------------------- collation over RCCL (no torch involved)
namespace {
// librccl is resolved at first use: the copy already in the process if there is one (torch links its own), else the system's
struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.h ? &api : nullptr;
  tried = true;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names)
    if (!api.h) api.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char* n : names)
    if (!api.h) api.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  if (!api.h) return nullptr;
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
  api.AllGather = (decltype(api.AllGather))dlsym(api.h, "ncclAllGather");
  api.AllReduce = (decltype(api.AllReduce))dlsym(api.h, "ncclAllReduce");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce || !api.GetErrorString) api.h = nullptr;
  return api.h ? &api : nullptr;
}
#define RCCLCHK(api, expr)                                                                                              \
  do {                                                                                                                  \
    ncclResult_t _r = (expr);                                                                                           \
    if (_r != ncclSuccess) return fail(WCA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(_r), __FILE__, __LINE__); \
  } while (0)
}  // namespace

int wca_comm_unique_id(uint8_t* id_out) {
  if (!id_out) return fail(WCA_ERR_INVALID, "null argument");
  RcclApi* r = rccl();
  if (!r) return fail(WCA_ERR_STATE, "librccl.so could not be loaded");
  static_assert(sizeof(ncclUniqueId) == WCA_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  RCCLCHK(r, r->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return WCA_OK;
}

int wca_comm_init(wca_engine* e, const uint8_t* id_in, int rank, int world) {
  if (!e || !id_in) return fail(WCA_ERR_INVALID, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(WCA_ERR_INVALID, "rank %d outside [0, %d)", rank, world);
  if (e->comm) return fail(WCA_ERR_STATE, "the engine already has a communicator (wca_comm_destroy first)");
  RcclApi* r = rccl();
  if (!r) return fail(WCA_ERR_STATE, "librccl.so could not be loaded");
  HIPCHK(hipSetDevice(e->device));
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  RCCLCHK(r, r->CommInitRank(&e->comm, world, id, rank));
  e->comm_rank = rank;
  e->comm_world = world;
  return WCA_OK;
}

int wca_comm_destroy(wca_engine* e) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  if (!e->comm) return WCA_OK;
  RcclApi* r = rccl();
  if (r) {
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    (void)r->CommDestroy(e->comm);
  }
  e->comm = nullptr;
  e->comm_world = 0;
  return WCA_OK;
}

int wca_collate_plan(const int64_t* sizes, const int64_t* capacities, int world, int64_t* pad_out) {
  if (!sizes || !capacities || world < 1) return fail(WCA_ERR_INVALID, "bad argument");
  int64_t mx = 0, min_cap = capacities[0];
  for (int i = 0; i < world; ++i) {
    if (sizes[i] < 0 || capacities[i] < 0) return fail(WCA_ERR_INVALID, "negative size");
    mx = sizes[i] > mx ? sizes[i] : mx;
    min_cap = capacities[i] < min_cap ? capacities[i] : min_cap;
  }
  if (pad_out) *pad_out = (mx + 15) / 16 * 16;
  if (mx > min_cap)
    return fail(WCA_ERR_TOO_LONG, "a rank packed %lld bytes, the smallest gather buffer holds %lld per rank (sizes are in sizes_host: retry)", (long long)mx,
                (long long)min_cap);
  return WCA_OK;
}

int wca_allgather_results(wca_engine* e, const uint8_t* packed_host, int64_t n_bytes, uint8_t* gathered_host, int64_t capacity_per_rank,
                          int64_t* sizes_host) {
  if (!e || !sizes_host || (n_bytes > 0 && !packed_host) || n_bytes < 0 || capacity_per_rank < 0) return fail(WCA_ERR_INVALID, "bad argument");
  if (!e->comm) return fail(WCA_ERR_STATE, "no communicator: call wca_comm_init first");
  RcclApi* r = rccl();
  HIPCHK(hipSetDevice(e->device));
  const int W = e->comm_world;
  // (1) every rank's {byte count, gather capacity}. The capacity is a per-caller argument, so "does it fit" must be decided on
  // what EVERY rank passed: all ranks then take the same branch and issue the same sequence of collectives
  const int64_t mine[2] = {n_bytes, capacity_per_rank};
  std::vector<int64_t> pairs(2 * (size_t)W), caps((size_t)W);
  HIPCHK(e->coll_send.ensure(sizeof(mine)));
  HIPCHK(e->coll_recv.ensure(sizeof(mine) * (size_t)W));
  HIPCHK(hipMemcpyAsync(e->coll_send.p, mine, sizeof(mine), hipMemcpyHostToDevice, e->stream));
  RCCLCHK(r, r->AllGather(e->coll_send.p, e->coll_recv.p, 2, ncclInt64, e->comm, e->stream));
  HIPCHK(hipMemcpyAsync(pairs.data(), e->coll_recv.p, sizeof(mine) * (size_t)W, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int i = 0; i < W; ++i) {
    sizes_host[i] = pairs[2 * i];
    caps[i] = pairs[2 * i + 1];
  }
  int64_t pad64 = 0;
  if (int rc = wca_collate_plan(sizes_host, caps.data(), W, &pad64)) return rc;  // identical inputs on every rank: identical verdict
  if (pad64 == 0) return WCA_OK;
  if (!gathered_host) return fail(WCA_ERR_INVALID, "null gather buffer");
  // (2) the packed records, padded to the largest count
  const size_t pad = (size_t)pad64;
  HIPCHK(e->coll_send.ensure(pad));
  HIPCHK(e->coll_recv.ensure(pad * (size_t)W));
  HIPCHK(hipMemsetAsync(e->coll_send.p, 0, pad, e->stream));
  if (n_bytes > 0) HIPCHK(hipMemcpyAsync(e->coll_send.p, packed_host, (size_t)n_bytes, hipMemcpyHostToDevice, e->stream));
  RCCLCHK(r, r->AllGather(e->coll_send.p, e->coll_recv.p, pad, ncclUint8, e->comm, e->stream));
  for (int i = 0; i < W; ++i)
    if (sizes_host[i] > 0)
      HIPCHK(hipMemcpyAsync(gathered_host + (size_t)i * (size_t)capacity_per_rank, (const char*)e->coll_recv.p + (size_t)i * pad, (size_t)sizes_host[i],
                            hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return WCA_OK;
}

int wca_allreduce_counters(wca_engine* e, int64_t* counters_host, int n) {
  if (!e || !counters_host || n < 1 || n > 64) return fail(WCA_ERR_INVALID, "bad argument");
  if (!e->comm) return fail(WCA_ERR_STATE, "no communicator: call wca_comm_init first");
  RcclApi* r = rccl();
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(e->coll_send.ensure(sizeof(int64_t) * 64));
  HIPCHK(hipMemcpyAsync(e->coll_send.p, counters_host, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, e->stream));
  RCCLCHK(r, r->AllReduce(e->coll_send.p, e->coll_send.p, (size_t)n, ncclInt64, ncclSum, e->comm, e->stream));
  HIPCHK(hipMemcpyAsync(counters_host, e->coll_send.p, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return WCA_OK;
}

// ---------------------------------------------------------------- kernel-level test entry points
int wca_test_gemm(wca_engine* e, const void* a, const void* w, const float* bias, void* c, int M, int N, int K, int gelu, int out_mode) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  HIPCHK(hipSetDevice(e->device));
  GemmArgs g{};
  g.A = (const half_t*)a;
  g.lda = K;
  g.W = (const half_t*)w;
  g.ldw = K;
  g.bias = bias;
  g.C = c;
  g.ldc = N;
  g.M = M;
  g.N = N;
  g.K = K;
  g.gelu = gelu;
  g.out_mode = out_mode & 0xff;
  if (g.out_mode == 4) {  // f16 pair output: c [M][2N], hi at column n, lo at column N + n
    g.ldc = 2 * N;
    g.c_lo = N;
  }
  g.force_tile = (out_mode >> 8) & 0xfff;  // 0 auto / 128 / 256 / 257 (persistent) / 258 (one tile per workgroup)
  g.supertile = out_mode >> 20;             // 0 = launch_gemm's choice (tools: tile-order experiments)
  g.sk_part = e->sk_big[0];                  // few tiles, K >= 2048, out_mode 2: split-K with the engine's workspace, as the encoder does
  g.sk_bytes = e->sk_big_bytes;
  HIPCHK(launch_gemm(g, e->stream));
  return WCA_OK;
}

int wca_test_gemm_pairs(wca_engine* e, const void* a2, const void* w, const float* bias, void* c, int M, int N, int K, int gelu, int out_mode) {
  if (!e || !a2 || !w || !c) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  const int om = out_mode & 0xff;
  if (!gemm_splitw_supported(M, N, K, 2 * K, om)) return fail(WCA_ERR_INVALID, "the pair-operand kernel does not take M=%d N=%d K=%d out_mode %d", M, N, K, om);
  GemmArgs g{};
  g.A = (const half_t*)a2;
  g.lda = 2 * K;
  g.a_lo = K;
  g.W = (const half_t*)w;
  g.ldw = K;
  g.bias = bias;
  g.C = c;
  g.ldc = N;
  g.M = M;
  g.N = N;
  g.K = K;
  g.gelu = gelu;
  g.out_mode = om;
  if (om == 4) {
    g.ldc = 2 * N;
    g.c_lo = N;
  }
  g.force_tile = (out_mode >> 8) & 0xfff;
  g.site = 1;
  HIPCHK(launch_gemm(g, e->stream));
  return WCA_OK;
}

int wca_test_gemm_ln(wca_engine* e, const void* a, const void* w, const float* bias, float* x, const float* gamma, const float* beta,
                     void* xn, int M, int N, int K, int site) {
  if (!e || !a || !w || !x || !gamma || !beta || !xn) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (!gemm_ln_supported(M, N, K, e->n_cu)) return fail(WCA_ERR_INVALID, "residual + LayerNorm epilogue not available for M=%d N=%d K=%d", M, N, K);
  const size_t mpad = align_up((size_t)M, 256);
  HIPCHK(e->tmp0.ensure(sizeof(unsigned long long) * (size_t)(N / 256) * mpad));
  HIPCHK(e->tmp1.ensure(sizeof(unsigned) * (mpad / 256 + 16)));
  HIPCHK(hipMemsetAsync(e->err_dev, 0, sizeof(int), e->stream));
  GemmArgs g{};
  g.A = (const half_t*)a;
  g.lda = K;
  g.W = (const half_t*)w;
  g.ldw = K;
  g.bias = bias;
  g.C = x;
  g.ldc = N;
  g.M = M;
  g.N = N;
  g.K = K;
  g.out_mode = 3;
  g.site = site;
  g.ln_gamma = gamma;
  g.ln_beta = beta;
  g.ln_out = (half_t*)xn;
  g.ln_ld = N;
  g.ln_eps = 1e-5f;
  g.ln_stats = (unsigned long long*)e->tmp0.p;
  g.ln_cnt = (unsigned*)e->tmp1.p;
  g.ln_err = e->err_dev;
  HIPCHK(launch_gemm(g, e->stream));
  HIPCHK(hipMemcpyAsync(e->err_host, e->err_dev, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->err_host[0] & 2) return fail(WCA_ERR_HIP, "LayerNorm statistics hand-off timed out");
  return WCA_OK;
}

int wca_test_gemm_rows(wca_engine* e, const void* a_f16, const float* x_f32, const float* gamma, const float* beta, const void* w,
                       const float* bias, void* c, int M, int N, int K, int gelu, int out_mode, int splitk, int groups, void* kv_k, void* kv_v,
                       int T_max, int kv_t) {
  if (!e || !w || !c || (!a_f16 && !x_f32) || (x_f32 && (!gamma || !beta))) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  if (splitk <= 0) splitk = gemm_rows_pick_splitk(K);
  if (splitk <= 0) return fail(WCA_ERR_INVALID, "no split of K=%d fits the few-row kernel", K);
  const size_t tiles = (size_t)((M + 63) / 64) * ((N + 15) / 16);
  if (splitk > 1) {
    HIPCHK(e->tmp0.ensure(gemm_rows_workspace_bytes(M, N, splitk)));
    HIPCHK(e->tmp1.ensure(sizeof(unsigned) * tiles));
    HIPCHK(hipMemsetAsync(e->tmp1.p, 0, sizeof(unsigned) * tiles, e->stream));
  }
  GemmArgs g{};
  g.A = (const half_t*)a_f16;
  g.lda = K;
  g.A32 = x_f32;
  g.lda32 = K;
  g.ln_gamma = gamma;
  g.ln_beta = beta;
  g.ln_eps = 1e-5f;
  g.W = (const half_t*)w;
  g.ldw = K;
  g.bias = bias;
  g.C = c;
  g.ldc = N;
  g.M = M;
  g.N = N;
  g.K = K;
  g.gelu = gelu;
  g.out_mode = out_mode;
  g.splitk = splitk;
  g.groups = groups;
  g.sk_part = (float*)e->tmp0.p;
  g.sk_cnt = (unsigned*)e->tmp1.p;
  g.kv_k = (half_t*)kv_k;
  g.kv_v = (half_t*)kv_v;
  g.kv_bs = (long)T_max * (N / 3);
  g.kv_t = kv_t;
  g.kv_d = kv_k ? N / 3 : 0;
  HIPCHK(launch_gemm_rows(g, e->stream));
  if (splitk > 1) {
    // the counters must be back at zero (self-cleaning): a second launch on the same workspace has to give the same result
    std::vector<unsigned> cnt(tiles);
    HIPCHK(hipMemcpyAsync(cnt.data(), e->tmp1.p, sizeof(unsigned) * tiles, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (unsigned v : cnt)
      if (v != 0) return fail(WCA_ERR_HIP, "split-K arrival counter left at %u", v);
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  return WCA_OK;
}

int wca_test_gemm_stamped(wca_engine* e, const void* a, const void* w, void* c, int M, int N, int K, int out_mode, unsigned long long* dbg_dev) {
  const int wrap_m = (out_mode >> 12) & 0xf, wrap_n = (out_mode >> 16) & 0xf;
  if (!e || (!dbg_dev && wrap_m == 0)) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  const bool pairs = (out_mode >> 9) & 1;
  GemmArgs g{};
  g.A = (const half_t*)a;
  g.lda = pairs ? 2 * K : K;
  g.a_lo = pairs ? K : 0;
  g.W = (const half_t*)w;
  g.ldw = K;
  g.C = c;
  g.ldc = N;
  g.M = M;
  g.N = N;
  g.K = K;
  g.out_mode = out_mode & 0xff;
  g.gelu = (out_mode >> 8) & 1;
  if (g.out_mode == 4) {
    g.ldc = 2 * N;
    g.c_lo = N;
  }
  g.site = 1;
  g.force_tile = pairs ? 0 : 257;
  g.dbg = dbg_dev;
  g.dbg_wrap_m = wrap_m;
  g.dbg_wrap_n = wrap_n;
  g.dbg_wrap_kind = (out_mode >> 20) & 7;
  HIPCHK(launch_gemm(g, e->stream));
  return WCA_OK;
}

int wca_test_attention_stamped(wca_engine* e, const void* q, const void* k, const void* v, void* o, int B, int H, int nq, int nk,
                               unsigned long long* dbg_dev);

int wca_test_set_attn_split_drop(int mask) {
  if (mask != 0 && mask != 1 && mask != 2 && mask != 3 && mask != 4 && mask != 8 && mask != 9 && mask != 12 && mask != 15)
    return fail(WCA_ERR_INVALID, "attention pass mask %d is not instantiated", mask);
  set_debug_switch("attn_split_drop", mask);
  return WCA_OK;
}

int wca_test_set_switch(const char* name, int value) {
  if (!name) return fail(WCA_ERR_INVALID, "null argument");
  if (set_debug_switch(name, value) != 0) return fail(WCA_ERR_INVALID, "unknown switch %s", name);
  return WCA_OK;
}

int wca_test_last_scores(wca_engine* e, int batch, float* scores_host) {
  if (!e || !scores_host) return fail(WCA_ERR_INVALID, "null argument");
  if (e->enq_count != e->fetch_count) return fail(WCA_ERR_STATE, "fetch the batches in flight first");
  HIPCHK(hipSetDevice(e->device));
  const size_t n = (size_t)batch * e->dims.n_text_layer * e->dims.n_text_head;
  if (batch < 1 || e->scores.bytes < n * sizeof(float)) return fail(WCA_ERR_INVALID, "no head scores of a batch of %d are held", batch);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipStreamSynchronize(e->stream2));
  HIPCHK(hipMemcpy(scores_host, e->scores.p, n * sizeof(float), hipMemcpyDeviceToHost));
  return WCA_OK;
}

static int test_attention_impl(wca_engine* e, const void* q, const void* k, const void* v, void* o, float* cap_dev, int cap_ld, int cap_cols,
                               int B, int H, int nq, int nk, int causal, unsigned long long* dbg) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  HIPCHK(hipSetDevice(e->device));
  AttnArgs a{};
  const int d = H * 64;
  a.Q = (const half_t*)q;
  a.q_bs = (long)nq * d;
  a.q_rs = d;
  a.K = (const half_t*)k;
  a.k_bs = (long)nk * d;
  a.k_rs = d;
  a.V = (const half_t*)v;
  a.v_bs = (long)nk * d;
  a.v_rs = d;
  a.O = (half_t*)o;
  a.o_bs = (long)nq * d;
  a.o_rs = d;
  a.cap = cap_dev;
  a.cap_bs = (long)H * nq * cap_ld;
  a.cap_hs = (long)nq * cap_ld;
  a.cap_ld = cap_ld;
  a.cap_cols = cap_cols;
  a.nq = nq;
  a.nk = nk;
  a.H = H;
  a.B = B;
  a.scale = 0.125f;
  a.causal = causal & 1;
  a.variant = (causal >> 8) & 3;  // 0 auto, 1 the 16x16x32 kernel, 2 the 32x32x16 kernel (attention.hip)
  a.dbg = dbg;
  HIPCHK(launch_attention(a, e->stream));
  return WCA_OK;
}

int wca_test_attention(wca_engine* e, const void* q, const void* k, const void* v, void* o, float* cap_dev, int cap_ld, int cap_cols,
                       int B, int H, int nq, int nk, int causal) {
  return test_attention_impl(e, q, k, v, o, cap_dev, cap_ld, cap_cols, B, H, nq, nk, causal, nullptr);
}

int wca_test_attention_split(wca_engine* e, const void* q2, const void* k2, const void* v2, void* o2, float* cap_dev, int cap_ld, int cap_cols,
                             int B, int H, int nq, int nk, int causal) {
  if (!e || !q2 || !k2 || !v2 || !o2) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  AttnArgs a{};
  const int d = H * 64;
  a.Q = (const half_t*)q2;
  a.q_bs = (long)nq * 2 * d;
  a.q_rs = 2 * d;
  a.K = (const half_t*)k2;
  a.k_bs = (long)nk * 2 * d;
  a.k_rs = 2 * d;
  a.V = (const half_t*)v2;
  a.v_bs = (long)nk * 2 * d;
  a.v_rs = 2 * d;
  a.O = (half_t*)o2;
  a.o_bs = (long)nq * 2 * d;
  a.o_rs = 2 * d;
  a.split = 1;
  a.q_lo = a.k_lo = a.v_lo = a.o_lo = d;
  a.cap = cap_dev;
  a.cap_bs = (long)H * nq * cap_ld;
  a.cap_hs = (long)nq * cap_ld;
  a.cap_ld = cap_ld;
  a.cap_cols = cap_cols;
  a.nq = nq;
  a.nk = nk;
  a.H = H;
  a.B = B;
  a.scale = 0.125f;
  a.causal = causal & 1;
  HIPCHK(launch_attention(a, e->stream));
  return WCA_OK;
}

int wca_test_attention_stamped(wca_engine* e, const void* q, const void* k, const void* v, void* o, int B, int H, int nq, int nk,
                               unsigned long long* dbg_dev) {
  if (!dbg_dev) return fail(WCA_ERR_INVALID, "null argument");
  return test_attention_impl(e, q, k, v, o, nullptr, 0, 0, B, H, nq >= 0 ? nq : ((-nq) & 0xfffff), nk, nq >= 0 ? 0 : (((-nq) >> 20) << 8), dbg_dev);  // nq < 0: -(variant << 20 | nq)
}

int wca_test_decode_select(wca_engine* e, const float* logits_dev, int batch, int n_vocab, int32_t* tokens_dev, int T_max, int cur_len,
                           int n_initial, const uint8_t* suppress_mask_dev, const uint8_t* blank_mask_dev, const wca_decode_opts* o,
                           float* sum_logprob_dev, int32_t* n_done_dev) {
  if (!e || !logits_dev || !tokens_dev || !suppress_mask_dev || !o || !sum_logprob_dev || !n_done_dev) return fail(WCA_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(e->device));
  DecodeSelectArgs a{};
  a.logits = logits_dev;
  a.ld = n_vocab;
  a.n_vocab = n_vocab;
  a.tokens = tokens_dev;
  a.T_max = T_max;
  a.cur_len = cur_len;
  a.n_initial = n_initial;
  a.suppress_mask = suppress_mask_dev;
  a.blank_mask = blank_mask_dev;
  a.eot = o->eot;
  a.timestamp_begin = o->timestamp_begin;
  a.apply_timestamp_rules = o->apply_timestamp_rules;
  a.max_initial_timestamp_index = o->max_initial_timestamp_index;
  a.sum_logprob = sum_logprob_dev;
  a.n_done = n_done_dev;
  HIPCHK(launch_decode_select(a, batch, e->stream));
  return WCA_OK;
}

int wca_test_layernorm(wca_engine* e, const float* x, const float* g, const float* b, void* out, int rows, int d) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(launch_layernorm_f16(x, g, b, (half_t*)out, rows, d, 1e-5f, e->stream));
  return WCA_OK;
}

int wca_test_layernorm_split(wca_engine* e, const float* x, const float* g, const float* b, void* out2, int rows, int d) {
  if (!e) return fail(WCA_ERR_INVALID, "null engine");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(launch_layernorm_f16(x, g, b, (half_t*)out2, rows, d, 1e-5f, e->stream, 2 * d, d));
  return WCA_OK;
}

int wca_test_encoder(wca_engine* e, const float* mel_dev, int batch, float* xa_out_dev) {
  int rc = check_ready(e);
  if (rc) return rc;
  if ((rc = join_phase2(e))) return rc;
  if (batch < 1 || batch > e->max_batch) return fail(WCA_ERR_INVALID, "batch %d outside [1,%d]", batch, e->max_batch);
  const wca_model_dims& D = e->dims;
  if ((rc = ensure_split_weights(e))) return rc;
  if ((rc = mel_to_tm(e, mel_dev, batch))) return rc;
  rc = run_encoder(e, batch);
  if (rc) return rc;
  // xn holds ln_post(x) in f16 (split mode: hi + lo pairs); widen for the caller
  const size_t n = (size_t)batch * N_CTX * D.n_audio_state;
  if (site_on(e, WCA_PSITE_CROSS_KV))  // ln_post wrote pairs for the cross-K/V projection
    hipLaunchKernelGGL(widen_split_kernel, dim3(2048), dim3(256), 0, e->stream, e->xn, xa_out_dev, (size_t)batch * N_CTX, D.n_audio_state);
  else
    hipLaunchKernelGGL(widen_kernel, dim3(2048), dim3(256), 0, e->stream, e->xn, xa_out_dev, n);
  HIPCHK(hipGetLastError());
  return WCA_OK;
}

}  // extern "C"
