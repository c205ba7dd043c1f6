#!/usr/bin/env python3
"""Throughput benchmark of the forced-alignment hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N == 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole per-utterance pipeline (log-mel -> Whisper encoder/decoder with every
cross-attention head captured -> median filter -> softmax -> head scores / top-k -> aggregation -> DTW ->
word times) over one micro-batch of synthetic utterances per GPU: whisper-medium dimensions with seeded
random weights (no checkpoint exists offline), 10 s of 16 kHz audio, 64-character teacher text, char
alignment, --aggr topk --topk 10 --medfilt_width 3 (BASELINE.json configs[1], TIMIT-shaped).
Utterances shard across ranks with no data-path collective; one RCCL all-gather collates the results.

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the dominant
kernel (encoder MLP fc1 GEMM) measured live with HIP events, and `cpu_baseline` (the oracle's PyTorch-CPU
restatement of the same pipeline, timed on this box's host cores on a bounded sample).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "utterances/sec (whisper-medium, 10s audio, char align) at 1/2/4/8 MI355X"
MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16/f16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed micro-batches per GPU (the two-deep pipeline drains once inside the timed region)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--model", type=str, default="medium")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--chars", type=int, default=64)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--medfilt_width", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-utts", type=int, default=4, help="utterances timed by the CPU baseline (after 1 warm-up): ~10-12 s of CPU work")
    ap.add_argument("--stages", action="store_true", help="print a per-stage HIP-event breakdown to stderr")
    return ap.parse_args()


def build_inputs(wca_pkg, syn, tok_mod, retok, args, n_batches, rank, world, device):
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
                            n_samples=[n_samples] * args.batch, max_frames=[n_samples // 320] * args.batch, texts=texts))
    return tok, batches


def cpu_baseline(args, sd, dims, syn, tok_mod, retok, audio_mod, oracle_times):
    """Oracle (CPU restatement of the reference pipeline, kind 'port') on a bounded sample."""
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    cores = min(os.cpu_count() or 1, 16)
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
        _words, st, en, _m, _s = timing_ref.force_align(w, tt, tok, "char", "topk", args.topk)
        dt = time.perf_counter() - t0
        oracle_times.append((10_000 + u, text, np.asarray(st), np.asarray(en)))
        if u > 0:
            times.append(dt)
    per = float(np.mean(times))
    return {"value": 1.0 / per, "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": "%d utterances (after 1 warm-up) of the same synthetic workload, batch 1 serial like infer_ali.py, "
                      "PyTorch-CPU fp32 forward + oracle post-processing, %.2f s/utt" % (len(times), per)}


def parity_against_oracle(args, model, tok, opts, timing, retok, syn, oracle_times, device):
    """The utterances the CPU baseline just aligned, through the GPU path inside a FULL bench-sized micro-batch (so the
    persistent GEMMs, the batched attention grid and the batched DTW run at exactly the timed configuration); compares
    word start / end times with the oracle's. Not part of the timed region."""
    n_samples = int(args.seconds * 16000)
    ids = [u for u, _t, _s, _e in oracle_times]
    ids = (ids * ((args.batch + len(ids) - 1) // len(ids)))[:args.batch]  # fill the batch by repetition
    pcm = np.stack([syn.synth_audio(u, n_samples) for u in ids])
    texts = {u: t for u, t, _s, _e in oracle_times}
    rows, tts = [], []
    for u in ids:
        tt = retok.encode(texts[u], tok, "char")
        tts.append(tt)
        rows.append([*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot])
    n_max = max(len(r) for r in rows)
    toks = np.full((len(rows), n_max), tok.eot, dtype=np.int64)
    for j, r in enumerate(rows):
        toks[j, :len(r)] = r
    jump, _ = model.align_batch(torch.from_numpy(pcm).to(device), [n_samples] * len(ids), torch.from_numpy(toks).to(device),
                                [len(r) for r in rows], [n_samples // 320] * len(ids), opts)
    total = within = identical = 0
    for j, u in enumerate(ids[:len(oracle_times)]):
        _w, st, en = timing.words_from_jump_frames(jump[j], tts[j], tok, "char")
        _u, _t, rst, ren = oracle_times[j]
        for a, b in ((np.asarray(st), rst), (np.asarray(en), ren)):
            if len(a) != len(b):
                return {"utterances": len(oracle_times), "error": "word count differs"}
            total += len(a)
            within += int(np.sum(np.abs(a - b) <= 0.02 + 1e-9))
            identical += int(np.sum(a == b))
    # batch invariance: the repeated copies of an utterance inside the batch must give the same frames
    invariant = all(np.array_equal(jump[j][:len(rows[j])], jump[j % len(oracle_times)][:len(rows[j])]) for j in range(len(ids)))
    return {"utterances": len(oracle_times), "word_boundaries": total, "within_one_frame": within, "identical": identical,
            "batch_invariant": bool(invariant), "tolerance": "one 20 ms encoder frame (north_star)"}


def measured_traffic(args, dims):
    """L2<->fabric bytes per launch of the dominant kernel from the rocprofv3 PMC passes recorded under profiles/
    (FETCH_SIZE doubled as the gfx950 note in MI355X_MICROARCH.md prescribes, + WRITE_SIZE); only quoted when this
    run uses the configuration those passes were collected on, else null."""
    path = os.path.join(ROOT, "profiles", "r01_dominant_kernel_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec["M"] == args.batch * 1500 and rec["N"] == 4 * dims.n_audio_state and rec["K"] == dims.n_audio_state:
            return rec["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("WCA_DIST_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
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

    dims = wca.dims_for(args.model)
    sd = syn.random_state_dict(dims, seed=0)
    model = wca.WhisperAMD(dims, device=str(device), max_batch=args.batch)
    model.load_state_dict(sd)
    tok, batches = build_inputs(wca, syn, tok_mod, retok, args, 2, rank, world, device)
    opts = model.make_opts(aggregation="topk", topk=args.topk, sot_len=len(tok.sot_sequence), medfilt_width=args.medfilt_width,
                           qk_scale=1.0)
    n_max = batches[0]["tokens"].shape[1]

    def enqueue(i):
        b = batches[i % len(batches)]
        model.align_batch(b["pcm"], b["n_samples"], b["tokens"], b["n_tok"], b["max_frames"], opts, enqueue_only=True)

    def finish(i, collect=None):
        b = batches[i % len(batches)]
        jump, _sel = model.fetch(args.batch, n_max, opts)
        # host tail: word-boundary merge + jump frames -> word start/end times (timing.py:105-113)
        for j in range(args.batch):
            _words, _st, _en = timing.words_from_jump_frames(jump[j], b["texts"][j], tok, "char")
        if collect is not None:
            collect.append(jump)

    def step(i, collect=None):
        enqueue(i)
        finish(i, collect)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    model.set_profiling(True)
    stage_acc = np.zeros(8)
    results = []
    t0 = time.perf_counter()
    # software pipeline of depth 2: the host tail of step i-1 runs while the GPU executes step i
    for i in range(args.steps):
        enqueue(i)
        if i > 0:
            finish(i - 1, results)
    finish(args.steps - 1, results)
    # HIP-event pairs around every launch of the dominant kernel were recorded on the engine stream during the
    # last timed step (they are re-recorded by each enqueue): 24 launches of the encoder fc1 GEMM
    dom_n, dom_ms, dom_flops = model.dominant_kernel_ms()
    if args.stages:
        stage_acc += np.array(model.last_stage_ms())
    # collate: one all-gather of the packed per-utterance jump frames (the only collective on the path)
    coll_dev = device if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")
    packed = torch.from_numpy(np.stack(results).astype(np.int32)).to(coll_dev)
    if dist is not None:
        gathered = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=coll_dev)
        dist.all_gather_into_tensor(gathered, packed)
        _ = gathered.cpu()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    model.set_profiling(False)
    if dist is not None:
        tmax = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        total_utts = world * args.batch * args.steps
        avg_ms = dom_ms / max(dom_n, 1)
        achieved = dom_flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        out = {
            "metric": METRIC, "value": total_utts / elapsed, "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "configs[1] shape (TIMIT-like): whisper-%s dims, seeded random weights, %.0f s @ 16 kHz gated noise, "
                                   "%d-char teacher text, char align, aggr=topk topk=%d medfilt_width=%d" %
                                   (args.model, args.seconds, args.chars, args.topk, args.medfilt_width),
                       "batch_per_gpu": args.batch, "utterances_per_step": world * args.batch, "parallelism": "dp%d (utterance shards)" % world},
            "roofline": {"bound": "mfma", "kernel": "gemm256p_f16_kernel<0, true, 1, false> (encoder MLP fc1, M=%d N=%d K=%d)" %
                                                     (args.batch * 1500, 4 * dims.n_audio_state, dims.n_audio_state),
                         "achieved": achieved, "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_F16_DENSE_PEAK_TFLOPS, "traffic": measured_traffic(args, dims),
                         "avg_launch_ms": avg_ms, "launches_timed": dom_n},
        }
        if args.stages:
            names = ["logmel", "encoder", "cross_kv", "decoder", "head_stats", "topk_aggregate", "dtw", "total"]
            print("stage ms/step (last step): " + ", ".join("%s=%.3f" % (n, v) for n, v in zip(names, stage_acc)), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            oracle_times = []
            out["cpu_baseline"] = cpu_baseline(args, sd, dims, syn, tok_mod, retok, audio_mod, oracle_times)
            # same utterances through the GPU path at the timed configuration, checked against the oracle's word times
            out["cpu_baseline"]["parity"] = parity_against_oracle(args, model, tok, opts, timing, retok, syn, oracle_times, device)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
