#!/usr/bin/env python3
"""Throughput benchmark of the forced-alignment hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N == 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole per-utterance pipeline (log-mel -> Whisper encoder/decoder with every
cross-attention head captured -> median filter -> softmax -> head scores / top-k -> aggregation -> DTW ->
word times) over one micro-batch of synthetic utterances per GPU: whisper-medium dimensions with seeded
random weights (no checkpoint exists offline; the cross-attention q/k weights are widened so the maps are PEAKY and
the parity leg is sensitive to operand rounding), 10 s of 16 kHz audio, 64-character teacher text, char alignment,
--aggr topk --topk 10 --medfilt_width 3 (BASELINE.json configs[1], TIMIT-shaped). 1 024 distinct utterances per GPU
are resident in HBM and cycled. Utterances shard across ranks with no data-path collective; the results are collated
by the PRODUCT's collation (shard.allgather_results: packed records, size gather + one all-gather; 3-counter
all-reduce) inside the timed region.

The forward runs in the engine's REFERENCE precision mode (wca_set_precision(WCA_PRECISION_REFERENCE = SPLIT): every stage on f16
(hi, lo) operand pairs -- the only site set that reproduces the fp32 CPU reference's word times with margin on the 301-utterance
parity leg, profiles/r04_precision_ablation.txt) -- that is `value`. The f16-operand fast mode of the same engine is timed right
after it and reported as the secondary object `f16_operating_point` (it misses the one-frame tolerance on ~1.5 % of the boundaries).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  * `roofline`: the encoder kernel with the LARGEST total time in the step (HIP-event pairs around every launch of every
    encoder kernel site, recorded live on the stream the kernels run on), `kernels`: the same figures for every site;
  * `cpu_baseline`: the oracle's PyTorch-CPU restatement of the same pipeline timed on this box's host cores on a bounded
    sample, and the parity of the GPU path against it at exactly the timed configuration.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "utterances/sec (whisper-medium, 10s audio, char align) at 1/2/4/8 MI355X"
MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16/f16
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E ~8 TB/s
SITE_SYMBOL = {
    "qkv": "gemm256p_f16_kernel<0, false, 1, 0, 0> (encoder QKV projection, N=3d K=d)",
    "attention": "attn32_kernel<false> (encoder self-attention 1500x1500, head_dim 64)",
    "out_proj": "gemm256p_f16_kernel<3, false, 1, 0, 0> (encoder attention out-projection + f32 residual read-modify-write + mlp_ln in the epilogue; N=d K=d; with --no-fuse-ln the <2, ...> kernel, mlp_ln a separate launch)",
    "fc1": "gemm256p_f16_kernel<0, true, 1, 0, 0> (encoder MLP fc1 + GELU, N=4d K=d)",
    "fc2": "gemm256p_f16_kernel<3, false, 4, 0, 0> (encoder MLP fc2 + f32 residual read-modify-write + the next attn_ln / ln_post in the epilogue; N=d K=4d; with --no-fuse-ln the <2, ...> kernel)",
    "ln1": "layernorm_f16_v4_kernel (attn_ln launches + ln_post)",
    "ln2": "layernorm_f16_v4_kernel (mlp_ln launches)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed micro-batches per GPU (>= 20 s at the default batch)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--distinct-batches", type=int, default=16, help="distinct micro-batches resident per GPU (16 x 64 = 1 024 utterances)")
    ap.add_argument("--model", type=str, default="medium")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--chars", type=int, default=64)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--medfilt_width", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-utts", type=int, default=32, help="utterances timed by the CPU baseline (after 1 warm-up)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = every core this process may use, measured; recorded in the line)")
    ap.add_argument("--aligned-utts", type=int, default=8, help="utterances of the second parity leg (alignment-like planted checkpoint: "
                    "synthetic.aligned_state_dict), aligned by the CPU oracle and by the GPU path in a full batch; 0 = skip")
    ap.add_argument("--stages", action="store_true", help="print a per-stage HIP-event breakdown to stderr")
    ap.add_argument("--fuse-ln", action="store_true", help="A/B, f16 sites only: LayerNorms inside the residual GEMMs' epilogues (wca_set_fuse_ln(1); needs the GPU to "
                    "itself, so it is NOT the engine's default and not what the headline runs)")
    ap.add_argument("--dec-unfused", action="store_true", help="decoder GEMMs on <= 256 rows as separate LayerNorm / GEMM launches (A/B of the few-row kernel; matters at small batch)")
    ap.add_argument("--no-overlap", action="store_true", help="phase 2 on the same stream as phase 1 (clean per-kernel rocprofv3 averages)")
    ap.add_argument("--precision", choices=("reference", "split", "f16"), default="reference",
                    help="reference (default, the contract line; 'split' names the same mode): every stage on f16 (hi, lo) operand pairs against "
                         "the exact f16 weights, three-pass attention -- the fp32 forward of timing.py:58 to fp32 summation noise; "
                         "f16: operands rounded to f16 once (fastest, misses the tolerance on ~1.5 %% of the boundaries)")
    ap.add_argument("--collate", choices=("torch", "abi"), default="torch",
                    help="collation of the per-rank results: torch = torch.distributed collectives (backend nccl = RCCL; the default), "
                         "abi = the C ABI's wca_allgather_results / wca_allreduce_counters (ncclAllGather from libwca.so; the communicator id "
                         "travels over the torch.distributed store)")
    ap.add_argument("--no-f16-leg", action="store_true", help="skip the f16-operand throughput + parity leg (`f16_operating_point`) that follows the contract run")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / collation rehearsal without a GPU: every rank fabricates its "
                    "shard's results instead of aligning (CPU tests of the --gpus N self-launch with WCA_DIST_BACKEND=gloo)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` run plainly (no launcher, WORLD_SIZE unset): start N ranks -- one process per GPU, through
    torch.distributed.run exactly as the driver's own command line would -- BEFORE this process touches a GPU, relay rank 0's
    JSON line and exit with the launcher's code. Fails loudly when fewer than N devices are visible."""
    import socket
    backend = os.environ.get("WCA_DIST_BACKEND", "nccl")
    if backend == "nccl" and not args.dry_run:
        have = torch.cuda.device_count()  # counts devices without initialising the runtime
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (one rank per GPU over RCCL)" % (args.gpus, have))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for raw in proc.stdout:
        txt = raw.strip()
        is_line = False
        if txt.startswith("{"):
            try:
                is_line = "metric" in json.loads(txt)
            except ValueError:
                pass
        if is_line:
            line = txt
        else:
            sys.stderr.write(raw)  # anything else the ranks print stays out of the one-line contract
    rc = proc.wait()
    if rc != 0:
        raise SystemExit(rc)
    if line is None:
        raise SystemExit("bench.py --gpus %d: the %d ranks finished without a result line" % (args.gpus, args.gpus))
    rec = json.loads(line)
    if rec.get("n_gpus") != args.gpus or rec.get("config", {}).get("dist_ranks") != args.gpus:
        raise SystemExit("bench.py --gpus %d: the process group reported %s ranks" % (args.gpus, rec.get("config", {}).get("dist_ranks")))
    print(line, flush=True)


def dry_run(args, dist, rank, world):
    """The N-rank harness without the engine: each rank fabricates the word times of its utterance shard, then runs the product's
    collation and the contract's barrier / max-over-ranks timing. Exercises launcher, rendezvous and collectives on CPU."""
    shard = importlib.import_module("whisper-char-alignment_amd.shard")
    results = {}
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        for j in range(args.batch):
            utt = (i * args.batch + j) * world + rank
            rng = np.random.default_rng(utt)
            st = np.sort(rng.integers(0, 500, size=int(rng.integers(1, 12)))) / 50.0
            results[utt] = (st, st + 0.02)
    cpu = torch.device("cpu")
    merged = shard.allgather_results(results, device=cpu)
    counters = shard.allreduce_counters(len(results), len(results), len(results), device=cpu)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    assert len(merged) == world * args.steps * args.batch and counters[0] == len(merged)
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        total = world * args.batch * args.steps
        print(json.dumps({"metric": METRIC, "value": total / elapsed, "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                          "data": "synthetic", "dry_run": True,
                          "config": {"workload": "DRY RUN: fabricated word times, no GPU work (harness rehearsal only; not a measurement)",
                                     "dist_ranks": dist.get_world_size() if dist is not None else 1,
                                     "dist_backend": dist.get_backend() if dist is not None else None, "collated_utterances": len(merged),
                                     "collective_calls": dict(shard.COLLECTIVE_CALLS)},
                          "roofline": None, "cpu_baseline": None}), flush=True)


def build_inputs(syn, tok_mod, retok, args, n_batches, rank, world, device):
    tok = tok_mod.get_tokenizer(True, language="English")
    n_samples = int(args.seconds * 16000)
    batches = []
    for bi in range(n_batches):
        pcm = np.zeros((args.batch, n_samples), dtype=np.float32)
        toks, ntoks, texts = [], [], []
        for j in range(args.batch):
            utt = (bi * args.batch + j) * world + rank  # utterance ids interleave across ranks (shard i % R == r)
            pcm[j] = syn.synth_audio(utt, n_samples)
            text = syn.synth_text(utt, args.chars)
            tt = retok.encode(text, tok, "char")
            full = [*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot]
            toks.append(full)
            ntoks.append(len(full))
            texts.append(tt)
        n_max = max(ntoks)
        tarr = np.full((args.batch, n_max), tok.eot, dtype=np.int64)
        for j, f in enumerate(toks):
            tarr[j, :len(f)] = f
        batches.append(dict(pcm=torch.from_numpy(pcm).to(device), tokens=torch.from_numpy(tarr).to(device), n_tok=ntoks,
                            n_samples=[n_samples] * args.batch, max_frames=[n_samples // 320] * args.batch, texts=texts, n_max=n_max))
    return tok, batches


def host_cores(override=0):
    """Cores this process may use, as MEASURED: the affinity mask capped by the cgroup CPU quota. `override` (--cpu-threads, or
    WCA_CPU_THREADS) replaces it and is recorded in the line. Returns (threads used, measured cores, override or None)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    ov = int(override) if override else int(os.environ.get("WCA_CPU_THREADS", "0"))
    return (max(1, ov) if ov > 0 else max(1, n)), max(1, n), (ov if ov > 0 else None)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, sd, dims, syn, audio_mod, oracle_times):
    """Oracle (CPU restatement of the reference pipeline, kind 'port') on a bounded sample, every host core."""
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    cores, cores_measured, cores_override = host_cores(args.cpu_threads)
    torch.set_num_threads(cores)
    tok = tokenizer_ref.CharTokenizer()
    ref = whisper_ref.WhisperRef(sd, dims)
    filt = audio_mod.mel_filters(dims.n_mels)
    n_samples = int(args.seconds * 16000)
    times = []
    oracle_times.clear()
    for u in range(args.cpu_utts + 1):
        pcm = torch.from_numpy(syn.synth_audio(10_000 + u, n_samples))
        text = syn.synth_text(10_000 + u, args.chars)
        t0 = time.perf_counter()
        mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(pcm), filt)
        tt = tokenizer_ref.encode_char(text, tok)
        tokens = torch.tensor([*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot])
        w, _ = timing_ref.get_attentions(mel, tokens, ref, n_samples // 320, args.medfilt_width, 1.0)
        _words, st, en, matrix, _s = timing_ref.force_align(w, tt, tok, "char", "topk", args.topk)
        dt = time.perf_counter() - t0
        # (outside the timed part) every head's selection score, for the parity leg's near-tie diagnosis
        _sel_all, all_scores = timing_ref.filter_attention(w, w.shape[0] * w.shape[1], 1, 1, 0)
        head_scores = {l * w.shape[1] + h: sc for sc, (l, h), _n in all_scores}
        oracle_times.append((10_000 + u, text, np.asarray(st), np.asarray(en), matrix, list(tt), head_scores))
        print("cpu baseline utterance %d/%d: %.2f s" % (u, args.cpu_utts, dt), file=sys.stderr, flush=True)  # progress (long, silent otherwise)
        if u > 0:
            times.append(dt)
    per = float(np.mean(times))
    return {"value": 1.0 / per, "unit": "utterances/s", "cores": cores, "cores_measured": cores_measured, "cores_measured_as": "min(sched_getaffinity, cgroup cpu.max quota)",
            "cpu_threads_override": cores_override, "host_cpu_count": os.cpu_count(), "cpu": cpu_model(), "kind": "port",
            "sample": "%d utterances (after 1 warm-up) of the same synthetic workload, batch 1 serial like infer_ali.py:48,57, "
                      "PyTorch-CPU fp32 forward (torch.set_num_threads(%d)) + oracle post-processing, %.2f s/utt" % (len(times), cores, per)}


SELECTION_TIE_REL = 1e-3     # relative score gap under which a head swap is attributed to operand rounding (measured GPU-vs-oracle
                             # score deviation: 6e-4 ... 1.6e-3, profiles/r02_parity_probe.txt)
SELECTION_TIE_REL_SPLIT = 1e-5   # the same in split mode: measured score deviation ~1e-6 relative (tests/test_split_gpu.py)
CONDITION_NOISE_REL_SPLIT = 2e-5  # ... and the matrix deviation in split mode (measured ~2e-6 relative)
CONDITION_NOISE_REL = 3e-3   # relative noise put on the ORACLE's own aggregated matrix by the conditioning test: the level by which the
                             # GPU's maps / matrices actually differ from the oracle's (2.7e-3 / 1.7e-3 relative, same file)


def oracle_is_ill_conditioned(matrix, tt, st, en, eps=CONDITION_NOISE_REL, trials=32):
    """Does the fp32 ORACLE's own alignment move by more than one frame when its aggregated matrix is perturbed by `eps`
    relative gaussian noise (the size of the measured GPU-vs-oracle difference)? For such an utterance "within one frame of the CPU path" is
    decided by last-bit luck in any reduced-precision forward; the parity leg reports those utterances separately."""
    from oracle import timing_ref, tokenizer_ref
    tok = tokenizer_ref.CharTokenizer()
    _w, word_tokens = tokenizer_ref.split_tokens_on_spaces(list(tt) + [tok.eot], tok, "char")
    rng = np.random.default_rng(12345)
    for _ in range(trials):
        noisy = matrix * (1.0 + eps * torch.from_numpy(rng.standard_normal(tuple(matrix.shape)).astype(np.float32)))
        ti, tj = timing_ref.dtw(-noisy)
        s2, e2 = timing_ref.jumps_to_times(ti, tj, word_tokens)
        if np.max(np.abs(s2 - st)) > 0.02 + 1e-9 or np.max(np.abs(e2 - en)) > 0.02 + 1e-9:
            return True
    return False


def parity_against_oracle(args, model, tok, opts, timing, retok, syn, oracle_times, device):
    """The utterances the CPU baseline just aligned, through the GPU path inside FULL bench-sized micro-batches (so the
    persistent GEMMs, the batched attention grid and the batched DTW run at exactly the timed configuration); compares
    word start / end times with the oracle's. Not part of the timed region."""
    n_samples = int(args.seconds * 16000)
    tie_rel = SELECTION_TIE_REL if args.precision == "f16" else SELECTION_TIE_REL_SPLIT      # ("reference" / "split": pair operands)
    noise_rel = CONDITION_NOISE_REL if args.precision == "f16" else CONDITION_NOISE_REL_SPLIT
    total = within = identical = 0
    utt_clean = utt_ill = utt_tie = utt_bad = off_well = utt_ill_all = 0
    offenders = []
    invariant = True
    for lo in range(0, len(oracle_times), args.batch):
        chunk = oracle_times[lo:lo + args.batch]
        ids = [c[0] for c in chunk]
        ids = (ids * ((args.batch + len(ids) - 1) // len(ids)))[:args.batch]  # fill the batch by repetition
        pcm = np.stack([syn.synth_audio(u, n_samples) for u in ids])
        texts = {c[0]: c[1] for c in chunk}
        rows, tts = [], []
        for u in ids:
            tt = retok.encode(texts[u], tok, "char")
            tts.append(tt)
            rows.append([*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot])
        n_max = max(len(r) for r in rows)
        toks = np.full((len(rows), n_max), tok.eot, dtype=np.int64)
        for j, r in enumerate(rows):
            toks[j, :len(r)] = r
        jump, sel = model.align_batch(torch.from_numpy(pcm).to(device), [n_samples] * len(ids), torch.from_numpy(toks).to(device),
                                      [len(r) for r in rows], [n_samples // 320] * len(ids), opts)
        for j in range(len(chunk)):
            _w, st, en = timing.words_from_jump_frames(jump[j], tts[j], tok, "char")
            _u, _t, rst, ren, rmatrix, rtt, rscores = chunk[j]
            off = 0
            for a, b in ((np.asarray(st), rst), (np.asarray(en), ren)):
                if len(a) != len(b):
                    return {"utterances": len(oracle_times), "error": "word count differs"}
                total += len(a)
                within += int(np.sum(np.abs(a - b) <= 0.02 + 1e-9))
                identical += int(np.sum(a == b))
                off += int(np.sum(np.abs(a - b) > 0.02 + 1e-9))
            ill = oracle_is_ill_conditioned(rmatrix, rtt, rst, ren, eps=noise_rel)
            utt_ill_all += int(ill)
            if off == 0:
                utt_clean += 1
                continue
            # did the two paths select the same heads? If not: how far from the cut are the swapped heads in the ORACLE's scores
            ranked = sorted(rscores.values())
            kth = ranked[-args.topk]
            o_heads = {h for h, sc in rscores.items() if sc >= kth}
            g_heads = {int(x) for x in sel[j][:args.topk]}
            swapped = sorted(o_heads ^ g_heads)
            gap = max((abs(rscores[h] - kth) / abs(kth) for h in swapped), default=0.0)
            near_tie = bool(swapped) and gap < tie_rel
            offenders.append({"utterance": int(_u), "boundaries_outside": off, "same_heads": not swapped,
                              "swapped_heads_rel_score_gap": float(gap), "oracle_path_moves_under_%.0e_noise" % noise_rel: bool(ill)})
            if near_tie:
                utt_tie += 1
            elif ill:
                utt_ill += 1
            else:
                utt_bad += 1
                off_well += off
        # batch invariance: the repeated copies of an utterance inside the batch must give the same frames
        invariant = invariant and all(np.array_equal(jump[j][:len(rows[j])], jump[j % len(chunk)][:len(rows[j])]) for j in range(len(ids)))
    return {"utterances": len(oracle_times), "word_boundaries": total, "within_one_frame": within, "identical": identical,
            "utterances_all_within": utt_clean, "utterances_ill_conditioned": utt_ill_all,
            "utterances_with_offenders_near_tied_selection": utt_tie,
            "near_tied_selection_means": "the GPU selected a different top-k head set and every swapped head's fp32-oracle score is within "
                                         "%.0e (relative) of the k-th score, i.e. below the measured GPU-vs-oracle score deviation of this "
                                         "precision mode (f16: ~1e-3, reference / split: ~1e-6)" % tie_rel,
            "offenders": offenders,
            "utterances_with_offenders_ill_conditioned": utt_ill,
            "utterances_with_offenders_well_conditioned": utt_bad, "offending_boundaries_in_well_conditioned_utterances": off_well,
            "ill_conditioned_means": "the fp32 oracle's OWN path moves by more than one frame when its aggregated matrix is perturbed by "
                                     "%.0e relative noise (32 seeded trials; the level of the measured GPU-vs-oracle matrix deviation in this "
                                     "precision mode): rounding decides such an utterance either way" % noise_rel,
            "precision": args.precision, "batch_invariant": bool(invariant), "weights": "peaky (cross_qk_std=0.08)", "tolerance": "one 20 ms encoder frame (north_star)"}


def parity_alignment_like(args, wca, dims, syn, audio_mod, tok_mod, retok, timing, device):
    """Second parity leg, not timed: the same pipeline on a checkpoint whose cross-attention looks like a trained Whisper's
    alignment heads (synthetic.aligned_state_dict: sharp monotonic ridges in 12 planted heads, separated head scores, words
    spread over the audio). With random weights the maps carry no alignment and a few utterances are decided by rounding
    (first leg); here the DTW is well conditioned and the selection unambiguous, so the GPU path must reproduce the fp32
    oracle exactly: same heads in the same order, identical word times."""
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    sd = syn.aligned_state_dict(dims, seed=0)
    model = wca.WhisperAMD(dims, device=str(device), max_batch=args.batch).load_state_dict(sd)
    model.set_precision(args.precision)
    ref = whisper_ref.WhisperRef(sd, dims)
    tok = tok_mod.get_tokenizer(True, language="English")
    rtok = tokenizer_ref.CharTokenizer()
    filt = audio_mod.mel_filters(dims.n_mels)
    n_samples = int(args.seconds * 16000)
    ids = [20_000 + u for u in range(args.aligned_utts)]
    fill = (ids * ((args.batch + len(ids) - 1) // len(ids)))[:args.batch]
    pcm = np.stack([syn.synth_audio(u, n_samples) for u in fill])
    texts = [syn.synth_text(u, args.chars) for u in fill]
    tts = [retok.encode(t, tok, "char") for t in texts]
    rows = [[*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot] for tt in tts]
    n_max = max(len(r) for r in rows)
    toks = np.full((len(rows), n_max), tok.eot, dtype=np.int64)
    for j, r in enumerate(rows):
        toks[j, :len(r)] = r
    opts = model.make_opts(aggregation="topk", topk=args.topk, sot_len=len(tok.sot_sequence), medfilt_width=args.medfilt_width, qk_scale=1.0)
    jump, sel = model.align_batch(torch.from_numpy(pcm).to(device), [n_samples] * len(fill), torch.from_numpy(toks).to(device),
                                  [len(r) for r in rows], [n_samples // 320] * len(fill), opts)
    H = dims.n_text_head
    total = within = identical = heads_same = 0
    span = []
    for j in range(len(ids)):
        mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(pcm[j])), filt)
        tt = tokenizer_ref.encode_char(texts[j], rtok)
        tokens = torch.tensor([*rtok.sot_sequence, rtok.no_timestamps, *tt, rtok.eot])
        w, _ = timing_ref.get_attentions(mel, tokens, ref, n_samples // 320, args.medfilt_width, 1.0)
        _words, rst, ren, _m, rscores = timing_ref.force_align(w, tt, rtok, "char", "topk", args.topk)
        _w, st, en = timing.words_from_jump_frames(jump[j], tts[j], tok, "char")
        for a, b in ((np.asarray(st), np.asarray(rst)), (np.asarray(en), np.asarray(ren))):
            total += len(a)
            within += int(np.sum(np.abs(a - b) <= 0.02 + 1e-9))
            identical += int(np.sum(a == b))
        heads_same += int([int(h) for h in sel[j][:args.topk]] == [l * H + h for _s, (l, h), _n in rscores])
        span.append(float(ren[-1] - rst[min(1, len(rst) - 1)]))
        print("alignment-like parity utterance %d/%d" % (j + 1, len(ids)), file=sys.stderr, flush=True)
    del model
    torch.cuda.empty_cache()
    return {"checkpoint": "synthetic.aligned_state_dict(seed=0): 12 planted alignment heads, ridge at 7 frames per token", "precision": args.precision,
            "utterances": len(ids),
            "word_boundaries": total, "within_one_frame": within, "identical": identical, "utterances_with_identical_head_selection": heads_same,
            "mean_span_of_aligned_words_s": float(np.mean(span)) if span else None}


def measured_traffic(args, dims, site):
    """HBM-side bytes per launch of the dominant kernel from the separate rocprofv3 --pmc passes recorded under
    profiles/ (tools/pmc_traffic.py: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE).
    Quoted only when that file was collected on this configuration and this kernel; carries its provenance."""
    if args.precision == "f16":
        files = {"fc1": "r03_traffic_fc1.json", "fc2": "r03_traffic_fc2.json", "qkv": "r03_traffic_qkv.json", "out_proj": "r03_traffic_out_proj.json"}
    else:   # the pair-operand kernels (collected in reference mode)
        files = {"fc1": "r05_traffic_fc1_reference.json", "fc2": "r05_traffic_fc2_reference.json", "qkv": "r05_traffic_qkv_reference.json",
                 "attention": "r05_traffic_attention_reference.json"}
    name = files.get(site)
    if name is not None and not os.path.exists(os.path.join(ROOT, "profiles", name)):
        name = name.replace("r05_", "r04_")   # (the attention kernel is round 4's: its record stays valid until re-collected)
    if name is None:
        return None, None
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            rec = json.load(f)
        if rec.get("site") == site and rec["batch"] == args.batch and rec["model"] == args.model and rec.get("precision", "f16") == args.precision:
            return rec["traffic_bytes_per_launch"], {"file": "profiles/" + name, "commit": rec.get("commit"),
                                                     "collected": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run"}
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def executed_tflop_per_utt(dims, args):
    """Algorithmic TFLOP (2 x MAC, SURVEY.md 8d formulas) of what the fused alignment path executes for one utterance."""
    S, d, ff, L = 1500, dims.n_audio_state, 4 * dims.n_audio_state, dims.n_audio_layer
    dt, Ld, V = dims.n_text_state, dims.n_text_layer, dims.n_vocab
    n = args.chars + 5
    enc = 2 * dims.n_mels * 3 * d * 3000 + 2 * d * 3 * d * S + L * (8 * S * d * d + 4 * S * S * d + 4 * S * d * ff)
    per_dec = 8 * n * dt * dt + 4 * n * n * dt + 4 * n * dt * dt + 4 * S * dt * dt + 4 * n * S * dt + 4 * n * dt * 4 * dt
    dec = Ld * per_dec
    # elided: last layer's cross value projection (2 S dt^2), its P.V (2 n S dt), cross out-projection (2 n dt^2), MLP (16 n dt^2 / 2 ...)
    elided = 2 * S * dt * dt + 2 * n * S * dt + 2 * n * dt * dt + 4 * n * dt * 4 * dt
    return (enc + dec - elided) / 1e12   # (the vocabulary projection, 2 n dt V, is not run at all on this path)


def source_tree_sha1():
    """sha1 over the product's sources (sorted relative paths + contents of bench.py, the package's .py files, csrc/ and include/):
    identifies the code on a box without .git; `python -c 'import bench; print(bench.source_tree_sha1())'` in a checkout reproduces it."""
    import hashlib
    h = hashlib.sha1()
    pkg = os.path.join(ROOT, "whisper-char-alignment_amd")
    files = [os.path.join(ROOT, "bench.py")]
    for base, exts in ((pkg, (".py",)), (os.path.join(pkg, "csrc"), (".hip", ".h", ".cpp")), (os.path.join(pkg, "dropin"), (".py",)),
                       (os.path.join(ROOT, "include"), (".h",))):
        for dirpath, _dirs, names in os.walk(base):
            if base == pkg and dirpath != pkg:
                continue   # (csrc / dropin are listed on their own; build/ and __pycache__ are not sources)
            files += [os.path.join(dirpath, n) for n in names if n.endswith(exts)]
    for f in sorted(set(files), key=lambda x: os.path.relpath(x, ROOT)):
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def git_head():
    if os.environ.get("WCA_COMMIT"):   # the GPU box's copy has no .git: the launching command passes the hash in
        return os.environ["WCA_COMMIT"]
    try:
        got = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except (OSError, subprocess.SubprocessError):
        got = ""
    return got or ("tree:" + source_tree_sha1())   # no .git (a gpurun box, the driver's box): the source-tree hash instead of null


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    dist = None
    # WCA_FORCE_DIST=1 with one rank: the RCCL collation path (process group, GPU all-gathers, barriers) exactly as at N > 1, so that
    # it can be exercised on a 1-GPU box (two ranks cannot share one device under RCCL)
    if world > 1 or os.environ.get("WCA_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        backend = os.environ.get("WCA_DIST_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on a 1-GPU box
        if backend == "nccl" and args.dry_run:
            raise SystemExit("--dry-run has no GPU: set WCA_DIST_BACKEND=gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if dist is not None and dist.get_world_size() != args.gpus:
        raise SystemExit("the process group has %d ranks, --gpus asked for %d" % (dist.get_world_size(), args.gpus))
    if args.dry_run:
        dry_run(args, dist, rank, world)
        if dist is not None:
            dist.destroy_process_group()
        return
    if os.environ.get("WCA_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    wca = importlib.import_module("whisper-char-alignment_amd")
    syn = importlib.import_module("whisper-char-alignment_amd.synthetic")
    tok_mod = importlib.import_module("whisper-char-alignment_amd.tokenizer")
    retok = importlib.import_module("whisper-char-alignment_amd.retokenize")
    timing = importlib.import_module("whisper-char-alignment_amd.timing")
    audio_mod = importlib.import_module("whisper-char-alignment_amd.audio")
    shard = importlib.import_module("whisper-char-alignment_amd.shard")

    dims = wca.dims_for(args.model)
    sd = syn.random_state_dict(dims, seed=0, cross_qk_std=0.08)
    model = wca.WhisperAMD(dims, device=str(device), max_batch=args.batch)
    model.load_state_dict(sd)
    if args.precision == "split":
        args.precision = "reference"   # one mode, two names
    model.set_precision(args.precision)
    if args.no_overlap:
        model.set_overlap(False)
    # The engine's DEFAULTS are what is timed. --fuse-ln is an A/B of the f16 mode only: LayerNorms inside the
    # residual GEMMs' epilogues, an in-launch hand-off between workgroups that needs the GPU to itself (include/wca.h, wca_set_fuse_ln)
    shared_device = world > max(torch.cuda.device_count(), 1)
    fuse_ln = bool(args.fuse_ln) and not shared_device
    if fuse_ln:
        model.set_fuse_ln(True)
    if args.dec_unfused:
        model.set_decode_mode(False, 1)
    coll_engine = None
    if args.collate == "abi" and dist is not None:
        # one RCCL communicator per engine, created through the C ABI; only its 128-byte id uses the launcher's side channel
        box = [wca.WhisperAMD.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        model.comm_init(box[0], rank, world)
        coll_engine = model
    tok, batches = build_inputs(syn, tok_mod, retok, args, max(2, args.distinct_batches), rank, world, device)
    opts = model.make_opts(aggregation="topk", topk=args.topk, sot_len=len(tok.sot_sequence), medfilt_width=args.medfilt_width,
                           qk_scale=1.0)

    def enqueue(i):
        b = batches[i % len(batches)]
        model.align_batch(b["pcm"], b["n_samples"], b["tokens"], b["n_tok"], b["max_frames"], opts, enqueue_only=True)

    def finish(i, collect=None):
        b = batches[i % len(batches)]
        jump, _sel = model.fetch(args.batch, b["n_max"], opts)
        # host tail: word-boundary merge + jump frames -> word start/end times (timing.py:105-113)
        for j in range(args.batch):
            _words, st, en = timing.words_from_jump_frames(jump[j], b["texts"][j], tok, "char", want_words=False)
            if collect is not None:
                collect[(i * args.batch + j) * world + rank] = (st, en)

    SITES = ("qkv", "attention", "out_proj", "fc1", "fc2", "ln1", "ln2")

    def timed_region(steps, warmup):
        """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides, the product's
        collation inside the bracket, max over ranks. Returns (elapsed s, per-site HIP-event sums, sampled steps, stage ms, #collated)."""
        for i in range(warmup):
            enqueue(i)
            finish(i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        model.set_profiling(True)
        results = {}
        site_acc = {s_: [0, 0.0, 0.0, 0.0] for s_ in SITES}   # launches, summed ms, flops / launch, bytes / launch
        sampled = 0

        def sample_sites():
            # HIP-event pairs around every launch of every encoder kernel of the batch enqueued last (recorded on the engine's
            # stream by each enqueue); reading them waits for that batch, so only every 8th step is sampled (a sampled step delays
            # the next enqueue by the read: ~0.2 % of the timed region)
            nonlocal sampled
            for s_ in SITES:
                n, ms, fl, by = model.kernel_ms(s_)
                site_acc[s_][0] += n
                site_acc[s_][1] += ms
                site_acc[s_][2], site_acc[s_][3] = fl, by
            sampled += 1

        t0 = time.perf_counter()
        # software pipeline of depth 2: the host tail of step i-1 runs while the GPU executes step i
        for i in range(steps):
            enqueue(i)
            if i > 0:
                finish(i - 1, results)
            if i % 8 == 7 and i + 1 < steps:
                sample_sites()
        finish(steps - 1, results)
        # collate exactly like infer_ali.py does: packed (index, n, starts, ends) records through one size gather + one
        # all-gather, and the 3-counter all-reduce (the only collectives on the path; no-ops for one rank)
        coll_dev = device if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")
        merged = shard.allgather_results(results, device=coll_dev, engine=coll_engine)
        counters = shard.allreduce_counters(len(results), len(results), len(results), device=coll_dev, engine=coll_engine)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        sample_sites()   # the last step (complete: its results were fetched)
        stage_ms = model.last_stage_ms() if args.stages else None
        model.set_profiling(False)
        assert len(merged) == world * steps * args.batch and counters[0] == len(merged)
        if dist is not None:
            tmax = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed, {s_: tuple(site_acc[s_]) for s_ in SITES}, sampled, stage_ms, len(merged)

    def kernel_table(sites, precision):
        sym = dict(SITE_SYMBOL)
        if precision != "f16" or not fuse_ln:   # separate LayerNorm launches: the residual GEMMs are the <2, ...> kernels
            for k_ in ("out_proj", "fc2"):
                sym[k_] = sym[k_].replace("gemm256p_f16_kernel<3,", "gemm256p_f16_kernel<2,").replace(" in the epilogue", " as a separate launch")
        if precision != "f16":
            for k_, v_ in list(sym.items()):
                if k_ == "attention":
                    sym[k_] = v_.replace("attn32_kernel<false>", "attn_split32_kernel<0>") + " [pair operands: three MFMA passes per product; the " \
                        "MFMA pipe executes 3x the algorithmic flops quoted]"
                elif "gemm256p" in v_:
                    sym[k_] = v_.replace("gemm256p_f16_kernel<0,", "gemm256p_f16_kernel<4,").replace(", 0, 0>", ", 0, 2>") \
                        + " [pair operands: A rows [hi | lo], every W K-tile staged once (SPLITW_MODE 2: three A slots + one W slot); the MFMA pipe executes 2x the algorithmic flops quoted]"
        kernels = {}
        for s_, (n, ms, fl, by) in sites.items():
            if n == 0:
                continue  # e.g. the LayerNorm sites when the LayerNorms run inside the GEMM epilogues
            avg = ms / max(n, 1)
            mfma = s_ not in ("ln1", "ln2")
            ach = (fl if mfma else by) / (avg * 1e-3) / (1e12 if mfma else 1e9) if avg > 0 else 0.0
            kernels[s_] = {"symbol": sym[s_], "launches": n, "avg_launch_ms": avg, "total_ms": ms, "bound": "mfma" if mfma else "hbm",
                           "achieved": ach, "unit": "TFLOP/s" if mfma else "GB/s", "frac": ach / (MFMA_F16_DENSE_PEAK_TFLOPS if mfma else HBM_PEAK_GBS)}
        return kernels, sym

    elapsed, sites, sampled_steps, stage_ms, n_collated = timed_region(args.steps, args.warmup)

    if rank == 0:
        total_utts = world * args.batch * args.steps
        kernels, sym = kernel_table(sites, args.precision)
        dom = max(kernels, key=lambda s_: kernels[s_]["total_ms"])
        traffic, traffic_src = measured_traffic(args, dims, dom)
        exec_mult = 1.0 if args.precision == "f16" else (3.0 if dom == "attention" else 2.0)
        d = dims.n_audio_state
        # FLOPs the timed path EXECUTES per utterance: SURVEY 8(d)'s 1.356 T minus what the fused path elides because nobody reads it
        # (the last decoder layer stops after its cross-attention capture: its value projection, P.V, cross out-projection and MLP,
        # the final LayerNorm and the vocabulary projection)
        executed_tflop = executed_tflop_per_utt(dims, args)
        out = {
            "metric": METRIC, "value": total_utts / elapsed, "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if args.precision == "f16" else "f16x2 (hi + lo operand pairs on the f16 MFMA pipe: fp32-equivalent operands, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "configs[1] shape (TIMIT-like): whisper-%s dims, seeded random weights (peaky cross-attention), %.0f s @ 16 kHz "
                                   "gated noise, %d-char teacher text, char align, aggr=topk topk=%d medfilt_width=%d; %d distinct utterances per GPU"
                                   % (args.model, args.seconds, args.chars, args.topk, args.medfilt_width, len(batches) * args.batch),
                       "precision": args.precision + (" (operands rounded to f16 once, fp32 accumulate: the fast mode, NOT the contract line)" if args.precision == "f16" else
                                                      " (wca_set_precision(WCA_PRECISION_REFERENCE = SPLIT): sites %s on (hi, lo) operand pairs -- pair GEMMs with every "
                                                      "W K-tile staged once, three-pass attention; 21 050 / 21 050 boundaries identical to the fp32 CPU oracle on the 1 033 fixture "
                                                      "utterances (tests/test_e2e_gpu.py::test_contract_mode_parity_1033_fixture_utterances), head scores within 1e-5, rms 1.5e-6 (profiles/r05_attn_pass_ablation.txt); achieved / frac count "
                                                      "ALGORITHMIC flops, the MFMA pipe executes 2x (GEMM) / 3x (attention) of them; the f16-operand mode of the same run is "
                                                      "under `f16_operating_point`)" % "+".join(model.precision_sites[0])),
                       "engine_defaults": "every engine setting is the shipped default except the precision mode named above (LayerNorms as separate launches"
                                          + (": --fuse-ln A/B ON" if fuse_ln else "") + ")",
                       "batch_per_gpu": args.batch, "utterances_per_step": world * args.batch, "parallelism": "dp%d (utterance shards)" % world,
                       "streams": "one (no overlap)" if args.no_overlap else "phase 1 / phase 2 overlapped on two streams",
                       "layernorm": ("fused into the residual GEMMs' epilogues where the consumer reads single f16 rows (--fuse-ln A/B: wca_set_fuse_ln(1))" if fuse_ln
                                     else "separate launches (engine default)"),
                       "collation": "shard.allgather_results + allreduce_counters (product path), inside the timed region; "
                                    + ("through the C ABI (wca_allgather_results / wca_allreduce_counters: ncclAllGather / ncclAllReduce from libwca.so)"
                                       if coll_engine is not None else "torch.distributed collectives" if dist is not None else "one rank: passthrough"),
                       "dist_ranks": dist.get_world_size() if dist is not None else 1,
                       "dist_backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if dist is not None else None,
                       "collated_utterances": n_collated, "collective_calls": dict(shard.COLLECTIVE_CALLS),
                       "pipeline_tflops": total_utts * executed_tflop / elapsed,
                       "pipeline_tflops_note": "executed algorithmic TFLOP per utterance %.4f (SURVEY 8d total 1.356 minus the elided tail of the last "
                                               "decoder layer and the unused logits)" % executed_tflop,
                       "commit": git_head()},
            "roofline": {"bound": "mfma", "kernel": "%s, M=%d d=%d" % (sym[dom], args.batch * 1500, d),
                         "selected_as": "largest total time of the encoder kernel sites over the sampled steps of the timed region (every 8th step + the last: %d steps)" % sampled_steps,
                         "achieved": kernels[dom]["achieved"], "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": kernels[dom]["frac"], "traffic": traffic, "traffic_source": traffic_src,
                         "executed_tflops": kernels[dom]["achieved"] * exec_mult, "executed_frac": kernels[dom]["frac"] * exec_mult,
                         "executed_note": ("f16 MFMA flops the kernel executes per algorithmic flop: %.0fx (pair operands = fp32-equivalent arithmetic; against the "
                                           "chip's native fp32 MFMA peak of 157.3 TFLOP/s the achieved algorithmic rate is %.1fx)" % (exec_mult, kernels[dom]["achieved"] / 157.3))
                                          if exec_mult > 1 else "1x (single f16 operands)",
                         "algorithmic_bytes": sites[dom][3], "algorithmic_flops": sites[dom][2],
                         "avg_launch_ms": kernels[dom]["avg_launch_ms"], "launches_timed": kernels[dom]["launches"]},
            "kernels": kernels,
        }
        if stage_ms is not None:
            names = ["logmel", "encoder", "cross_kv", "decoder", "head_stats", "topk_aggregate", "dtw", "total"]
            print("stage ms/step (last step): " + ", ".join("%s=%.3f" % (n, v) for n, v in zip(names, stage_ms)), file=sys.stderr)
            print("encoder kernel sites (%d sampled steps of the timed region): " % sampled_steps + ", ".join("%s=%.3f ms x%d (%.0f %s)" % (
                s_, k["avg_launch_ms"], k["launches"], k["achieved"], k["unit"]) for s_, k in kernels.items()), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            oracle_times = []
            out["cpu_baseline"] = cpu_baseline(args, sd, dims, syn, audio_mod, oracle_times)
            # same utterances through the GPU path at the timed configuration, checked against the oracle's word times
            out["cpu_baseline"]["parity"] = parity_against_oracle(args, model, tok, opts, timing, retok, syn, oracle_times, device)
            if args.precision != "f16" and not args.no_f16_leg:
                # the SAME engine switched to the f16-operand fast mode: its throughput (a shorter timed region of the same shape) and
                # its parity against the SAME oracle results -- a secondary operating point, not the contract line
                contract_mode = args.precision
                model.set_precision("f16")
                steps2 = max(8, args.steps // 2)
                el2, sites2, _n2, _st2, _c2 = timed_region(steps2, 2)
                k2, _sym2 = kernel_table(sites2, "f16")
                args.precision = "f16"
                par2 = parity_against_oracle(args, model, tok, opts, timing, retok, syn, oracle_times, device)
                args.precision = contract_mode
                model.set_precision(contract_mode)
                out["f16_operating_point"] = {
                    "mode": "wca_set_precision(WCA_PRECISION_F16): GEMM / attention operands rounded to f16 once, fp32 accumulation -- narrower than the "
                            "reference's fp32 forward (timing.py:58); faster, and outside north_star's one-frame tolerance on the boundaries counted in `parity`",
                    "value": args.batch * steps2 / el2, "unit": "utterances/s", "steps": steps2, "ms_per_step": 1e3 * el2 / steps2,
                    "kernels": {s_: {"avg_launch_ms": v["avg_launch_ms"], "achieved": v["achieved"], "unit": v["unit"], "frac": v["frac"]} for s_, v in k2.items()},
                    "parity": par2}
            if args.aligned_utts > 0:
                out["cpu_baseline"]["parity_alignment_like"] = parity_alignment_like(args, wca, dims, syn, audio_mod, tok_mod, retok, timing, device)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
