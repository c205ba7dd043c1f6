#!/usr/bin/env python3
"""Forced-alignment driver with the reference's command line (infer_ali.py:151-173), running on the
MI355X engine. Same flags, same result JSON ({args..., precision, recall, f1, r_value}) and the same
`-predictions.pkl` schema (infer_ali.py:118-119,139-148), so eval_ali.py keeps working.

    python infer_ali.py --dataset TIMIT --scp scp/test.wav.scp --model medium --weights /path/medium.pt \
        --vocab /path/multilingual.tiktoken --aggr topk --topk 10 --aligned_unit_type char --medfilt_width 3 \
        --output_dir results/timit
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 infer_ali.py ...      (one rank per GPU)

Additions: --weights (LOCAL openai-format checkpoint; nothing is fetched by name), --random_init,
--batch_size (utterances per micro-batch through the fused wca_align_batch path), --vocab (local
tiktoken file: ASR text, subword mode), --teacher, --readers.
--teacher asr (the default, = the reference: greedy whisper.decode pre-pass, infer_ali.py:60-68, then teacher-forcing
of its hypothesis; the encoder runs ONCE per utterance, the reference runs it twice) needs --vocab to turn token
ids into text. --teacher text teacher-forces the dataset transcript instead and skips the pre-pass: that is NOT the
reference's flow (it gives optimistic scores) and is recorded as such in the printed and dumped results.

Host pipeline (the GPU never waits on Python I/O): a pool of reader threads decodes audio files (SPHERE / WAV / FLAC),
parses the ground truth and tokenises ahead of the GPU; micro-batches are staged in pinned memory and uploaded
asynchronously; up to two micro-batches are in flight in the engine (wca_align_batch_enqueue / _fetch), so the host
tail (word merge, times, scoring) of batch i-1 overlaps batch i on the GPU.
Utterances with max_frames > 1500 or more than 448 tokens are skipped and their id printed
(infer_ali.py:78-81).
"""
import argparse
import collections
import datetime
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

if __package__ in (None, ""):
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _pkg = importlib.import_module("whisper-char-alignment_amd")
    __package__ = _pkg.__name__

from . import shard as _shard  # noqa: E402
from .audio import N_SAMPLES_PER_TOKEN as AUDIO_SAMPLES_PER_TOKEN  # noqa: E402
from .dataset import AMI, TIMIT, LibriSpeech  # noqa: E402
from .engine import WhisperAMD, dims_for, MAX_FRAMES, MAX_LENGTH  # noqa: E402
from .metrics import eval_n1, eval_n1_strict, get_seg_metrics  # noqa: E402
from .retokenize import encode, remove_punctuation  # noqa: E402
from .timing import words_from_jump_frames, default_find_alignment  # noqa: E402
from .audio import log_mel_spectrogram, pad_or_trim  # noqa: E402
from .tokenizer import get_tokenizer  # noqa: E402
from .decoding import DecodingOptions, decode  # noqa: E402

DATASET = {"TIMIT": TIMIT, "LibriSpeech": LibriSpeech, "AMI": AMI}


def load_model(args, device):
    if args.weights:
        # like whisper.load_model(args.model): the model's curated alignment heads are installed (default_find_alignment)
        return WhisperAMD.from_checkpoint(args.weights, device=device, max_batch=args.batch_size, name=args.model)
    if args.random_init:
        from .synthetic import random_state_dict
        dims = dims_for(args.model)
        model = WhisperAMD(dims, device=device, max_batch=args.batch_size).load_state_dict(random_state_dict(dims, seed=0))
        model.use_official_alignment_heads(args.model)
        return model
    raise SystemExit("no weights: pass --weights /local/path/%s.pt (openai-whisper checkpoint; nothing is downloaded by name) "
                     "or --random_init for a dry run" % args.model)


def prefetch(dataset, indices, prepare, workers, depth):
    """Yields prepare(dataset.read(n), n) for n in indices, in order, with up to `depth` items being read / decoded /
    tokenised by `workers` threads ahead of the consumer (file reads, numpy and the C FLAC decoder release the GIL)."""
    def job(n):
        return prepare(n, dataset.read(n))
    with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
        window = collections.deque()
        it = iter(indices)
        for n in it:
            window.append(pool.submit(job, n))
            if len(window) >= depth:
                break
        while window:
            fut = window.popleft()
            nxt = next(it, None)
            if nxt is not None:
                window.append(pool.submit(job, nxt))
            yield fut.result()


class PinnedStager:
    """Two pinned host buffers for the micro-batch PCM; each upload is asynchronous and a buffer is re-used only after
    the copy that read it has completed (event)."""

    def __init__(self, device):
        self.device = device
        self.bufs = [None, None]
        self.events = [None, None]
        self.i = 0

    def upload(self, batch):
        smax = max(len(b["pcm"]) for b in batch)
        need = len(batch) * smax
        k = self.i
        self.i ^= 1
        if self.events[k] is not None:
            self.events[k].synchronize()
        if self.bufs[k] is None or self.bufs[k].numel() < need:
            self.bufs[k] = torch.empty(need, dtype=torch.float32).pin_memory()
        host = self.bufs[k][:need].view(len(batch), smax)
        hv = host.numpy()
        for j, b in enumerate(batch):
            n = len(b["pcm"])
            hv[j, :n] = b["pcm"]
            hv[j, n:] = 0.0
        dev = host.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return dev, [len(b["pcm"]) for b in batch]


def infer_dataset(args, model=None):
    """`model`: an already constructed WhisperAMD engine (tests / long-lived services); default: load_model(args)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)
    if args.teacher is None:
        # the reference always aligns the ASR hypothesis (infer_ali.py:60-68); that needs a vocabulary to read it
        if args.vocab is None and not args.random_init:
            raise SystemExit("the reference's flow (--teacher asr, the default) turns the greedy decode's token ids back into text: "
                             "pass --vocab <local multilingual.tiktoken>, or choose --teacher text explicitly to teacher-force the "
                             "dataset transcript (not the reference's behaviour)")
        args.teacher = "asr"
    if rank == 0:
        print(args)
        if args.teacher == "text":
            print("NOTE: --teacher text teacher-forces the GROUND-TRUTH transcript; the reference aligns the ASR hypothesis "
                  "(infer_ali.py:60-68), so these scores are optimistic w.r.t. the reference's protocol", file=sys.stderr)
        if args.plot:
            print("WARNING: --plot is not supported by this engine (no matplotlib side-car); ignored", file=sys.stderr)
    if model is None:
        model = load_model(args, device)
    if args.batch_size > model.max_batch:
        raise SystemExit("--batch_size %d exceeds the engine's max_batch %d" % (args.batch_size, model.max_batch))
    if getattr(args, "forward_precision", None) and model.precision != {"reference": "split"}.get(args.forward_precision, args.forward_precision):   # (`precision` is the P of P/R/F1 in the results)
        model.set_precision(args.forward_precision)   # "reference" / "split": the reference's fp32 forward to fp32 summation noise (wca_set_precision)
    if args.n_mels != model.dims.n_mels:
        raise SystemExit("--n_mels %d does not match the checkpoint (%d); large-v3 needs --n_mels 128" % (args.n_mels, model.dims.n_mels))
    tokenizer = get_tokenizer(model.is_multilingual, language="English", vocab_path=args.vocab)
    # language="en" + task=transcribe -> ASR and alignment using whisper (infer_ali.py:40)
    asr_options = DecodingOptions(language="en", vocab_path=args.vocab)
    if args.teacher == "asr" and args.vocab is None and not args.random_init:
        raise SystemExit("--teacher asr turns token ids back into text: pass --vocab <local multilingual.tiktoken>")
    extra = {"alignment_file": args.alignment_file} if args.alignment_file else {}
    if extra and args.dataset == "TIMIT":
        raise SystemExit("--alignment_file applies to LibriSpeech (ls_alignment_*.txt) and AMI (ami_kaldi.pkl)")
    dataset = DATASET[args.dataset](args.scp, n_mels=args.n_mels, device=device, model=model, compute_mel=False, **extra)
    mine = _shard.shard_indices(len(dataset), rank, world, [dataset.duration_hint(i) for i in range(len(dataset))])
    opts = model.make_opts(aggregation=args.aggr, topk=args.topk, w_colnorm=args.w_colnorm, w_rownorm=args.w_rownorm,
                           w_coverage=args.w_coverage, sot_len=len(tokenizer.sot_sequence), medfilt_width=args.medfilt_width,
                           qk_scale=1.0)
    corrects = total_preds = total_gts = 0
    all_predictions = {}
    local_times = {}
    stager = PinnedStager(device)

    def prepare(n, item):
        """Reader-thread side: text normalisation + tokenisation + the skip rule of infer_ali.py:78-81."""
        pcm, duration, texts, starts, ends, fid = item
        texts = remove_punctuation(texts)
        max_frames = duration // AUDIO_SAMPLES_PER_TOKEN
        if args.teacher == "text":
            text_tokens = encode(texts, tokenizer, args.aligned_unit_type)  # the dataset transcript is teacher-forced
            tokens = [*tokenizer.sot_sequence, tokenizer.no_timestamps, *text_tokens, tokenizer.eot]
        else:
            text_tokens, tokens = None, []  # filled from the ASR hypothesis when the micro-batch is decoded
        skip = max_frames > MAX_FRAMES or len(tokens) > MAX_LENGTH or max_frames < 1
        return dict(index=n, pcm=pcm, tokens=tokens, text_tokens=text_tokens, max_frames=int(max_frames), texts=texts,
                    starts=starts, ends=ends, fid=fid, skip_early=skip)

    def score(b, words, start_times, end_times):
        """infer_ali.py:114-132 for one utterance."""
        nonlocal corrects, total_preds, total_gts
        ends_hat = end_times
        local_times[b["index"]] = (np.asarray(start_times, dtype=np.float64), np.asarray(end_times, dtype=np.float64))
        if args.save_prediction:
            all_predictions[b["index"]] = dict(starts=b["starts"], ends=b["ends"], texts=b["texts"].split(), starts_hat=start_times,
                                               ends_hat=ends_hat, predwords=words, fids=b["fid"])
        if not args.strict:
            c, _ = eval_n1(b["ends"], ends_hat, args.tolerance)
            total_gts += len(b["ends"])
            total_preds += len(ends_hat)
            corrects += c
        else:
            hyp = " ".join(words[:-1]).split() if words else []
            tp, fp, fn = eval_n1_strict(b["ends"], ends_hat, b["texts"].split(), hyp, args.tolerance)
            corrects += tp
            total_gts += tp + fn
            total_preds += tp + fp

    def start_asr(batch):
        """--teacher asr, stage 1: enqueue log-mel + encoder + cross-K/V of this micro-batch (no host sync). It runs on the
        engine's first stream while the PREVIOUS micro-batch is decoded and aligned on the second one."""
        pcm_dev, n_samples = stager.upload(batch)
        model.encode_batch(pcm=pcm_dev, n_samples=n_samples)
        return batch

    def token_matrix(batch):
        n_max = max(len(b["tokens"]) for b in batch)
        toks = np.full((len(batch), n_max), tokenizer.eot, dtype=np.int64)
        for j, b in enumerate(batch):
            toks[j, :len(b["tokens"])] = b["tokens"]
        return torch.from_numpy(toks).to(device, non_blocking=True), n_max

    def finish(entry):
        """Host tail of a micro-batch whose GPU work was enqueued earlier: fetch the jump frames, merge words, score."""
        batch, n_max, _keep = entry
        jump, _ = model.fetch(len(batch), n_max, opts)
        for j, b in enumerate(batch):
            if b.get("skip"):
                continue
            words, start_times, end_times = words_from_jump_frames(jump[j], b["text_tokens"], tokenizer, args.aligned_unit_type)
            score(b, words, start_times, end_times)

    enqueued = collections.deque()  # micro-batches in flight in the engine (at most 2)

    def enqueue_text(batch):
        """--teacher text: upload + enqueue the fused pipeline; the host tail of the PREVIOUS batch runs meanwhile."""
        if not batch:
            return
        pcm_dev, n_samples = stager.upload(batch)
        toks_dev, n_max = token_matrix(batch)
        model.align_batch(pcm_dev, n_samples, toks_dev, [len(b["tokens"]) for b in batch], [b["max_frames"] for b in batch], opts,
                          enqueue_only=True)
        enqueued.append((batch, n_max, (pcm_dev, toks_dev)))  # the device buffers stay alive until the fetch
        while len(enqueued) > 1:
            finish(enqueued.popleft())

    def flush_asr(batch):
        """--teacher asr, stage 2 for a micro-batch whose encoder state is queued in the engine: greedy decode
        (infer_ali.py:60-61), tokenise the hypothesis, align re-using the encoder output, host tail."""
        if not batch:
            return
        results = decode(model, None, asr_options, encoded_batch=len(batch))
        for b, r in zip(batch, results):
            transcription = remove_punctuation(r.text)  # infer_ali.py:64
            try:
                b["text_tokens"] = encode(transcription, tokenizer, args.aligned_unit_type)
            except Exception as exc:  # non-ASCII hypothesis without a vocabulary file
                print("%s: cannot tokenize the ASR hypothesis (%s)" % (b["fid"], exc))
                b["text_tokens"] = None
            if b["text_tokens"] is not None:
                b["tokens"] = [*tokenizer.sot_sequence, tokenizer.no_timestamps, *b["text_tokens"], tokenizer.eot]
            if b["text_tokens"] is None or len(b["tokens"]) > MAX_LENGTH:
                print(b["fid"])  # infer_ali.py:79-81; the row stays in the batch (its encoder state is in place) but is not scored
                b["skip"] = True
                b["text_tokens"] = []
                b["tokens"] = [*tokenizer.sot_sequence, tokenizer.no_timestamps, tokenizer.eot]
        toks_dev, n_max = token_matrix(batch)
        model.align_batch(None, None, toks_dev, [len(b["tokens"]) for b in batch], [b["max_frames"] for b in batch], opts, enqueue_only=True)
        finish((batch, n_max, toks_dev))

    pending, in_flight = [], []
    n_done = 0
    t0 = time.time()
    for b in prefetch(dataset, mine, prepare, args.readers, max(2 * args.batch_size, 8)):
        if b["skip_early"]:
            print(b["fid"])
            continue
        if args.default_whisper_timing:  # per-utterance path (timing.py:116-186), not the fused batch path
            mel = log_mel_spectrogram(pad_or_trim(torch.from_numpy(b["pcm"])), args.n_mels, model=model)
            text_tokens = b["text_tokens"]
            if args.teacher == "asr":
                transcription = remove_punctuation(decode(model, mel, asr_options).text)
                text_tokens = encode(transcription, tokenizer, args.aligned_unit_type)
                if len(text_tokens) + len(tokenizer.sot_sequence) + 2 > MAX_LENGTH:
                    print(b["fid"])
                    continue
            words, start_times, end_times, _ws, _ = default_find_alignment(model, tokenizer, text_tokens, mel, b["max_frames"],
                                                                         medfilt_width=args.medfilt_width)
            score(b, words, start_times, end_times)  # honours --strict / --save_prediction like infer_ali.py:114-132
            n_done += 1
            continue
        pending.append(b)
        n_done += 1
        if len(pending) == args.batch_size:
            if args.teacher == "asr":
                # two-deep pipeline: encode this micro-batch, then decode + align the previous one beside it
                start_asr(pending)
                flush_asr(in_flight)
                in_flight = pending
            else:
                enqueue_text(pending)
            pending = []
    if args.teacher == "asr" and not args.default_whisper_timing:
        if pending:
            start_asr(pending)
        flush_asr(in_flight)
        flush_asr(pending)
    else:
        enqueue_text(pending)
        while enqueued:
            finish(enqueued.popleft())
    elapsed = time.time() - t0

    corrects, total_preds, total_gts = _shard.allreduce_counters(corrects, total_preds, total_gts)
    all_times = _shard.allgather_results(local_times)
    if args.save_prediction:
        all_predictions = _shard.gather_predictions(all_predictions)  # every rank's dict on rank 0 (None elsewhere)
    if rank == 0:
        precision, recall, f1, r_value, _ = get_seg_metrics(corrects, corrects, total_preds, total_gts)
        results = dict(precision=precision, recall=recall, f1=f1, r_value=r_value)
        print(results)
        print("aligned %d utterances on %d GPU(s) in %.1f s (%.1f utt/s)" % (len(all_times), world, elapsed, len(all_times) / max(elapsed, 1e-9)))
        filename = datetime.datetime.fromtimestamp(time.time()).strftime("%Y-%m-%d-%H:%M:%S")
        os.makedirs(args.output_dir, exist_ok=True)
        notes = {}
        if args.teacher == "text":
            notes["teacher_note"] = "ground-truth transcript teacher-forced; the reference aligns the ASR hypothesis (infer_ali.py:60-68)"
        with open(os.path.join(args.output_dir, filename + ".json"), "w") as f:
            json.dump({**vars(args), **results, **notes, "utterances": len(all_times), "seconds": elapsed}, f)
        if args.save_prediction:
            import joblib
            joblib.dump(all_predictions, os.path.join(args.output_dir, filename + "-predictions.pkl"))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return dict(utterances=len(all_times), seconds=elapsed)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Arguments for whisper-based forced alignments")
    p.add_argument("--model", type=str, default="medium")
    p.add_argument("--dataset", type=str, default="TIMIT", choices=["TIMIT", "LibriSpeech", "AMI"])
    p.add_argument("--scp", type=str, default="scp/test.wav.scp")
    p.add_argument("--output_dir", type=str, default="results", help="Path to the output directory", required=True)
    p.add_argument("--n_mels", type=int, default=80)
    p.add_argument("--medfilt_width", type=int, default=7)
    p.add_argument("--aggr", type=str, default="mean", choices=["mean", "topk"])
    p.add_argument("--topk", type=int, default=15)
    p.add_argument("--aligned_unit_type", type=str, default="subword", choices=["subword", "char"])
    p.add_argument("--tolerance", type=float, default=0.02)
    p.add_argument("--w_colnorm", type=float, default=1.0)
    p.add_argument("--w_rownorm", type=float, default=1.0)
    p.add_argument("--w_coverage", type=float, default=0.0)
    p.add_argument("--plot", action="store_true")
    p.add_argument("--strict", action="store_true")
    p.add_argument("--save_prediction", action="store_true")
    p.add_argument("--default_whisper_timing", action="store_true")
    # engine-specific additions
    p.add_argument("--weights", type=str, default=None, help="local openai-whisper checkpoint (.pt)")
    p.add_argument("--random_init", action="store_true", help="seeded random weights (dry run without a checkpoint)")
    p.add_argument("--vocab", type=str, default=None, help="local tiktoken vocabulary file (subword mode / non-ASCII text)")
    p.add_argument("--batch_size", type=int, default=16, help="utterances per micro-batch on each GPU")
    p.add_argument("--alignment_file", type=str, default=None, help="LibriSpeech ls_alignment_<split>.txt / AMI ami_kaldi.pkl")
    p.add_argument("--teacher", type=str, default=None, choices=["text", "asr"],
                   help="asr (default; needs --vocab): greedy decode pre-pass gives the teacher text, the reference's behaviour; "
                        "text: teacher-force the dataset transcript (not the reference's protocol)")
    p.add_argument("--forward_precision", type=str, default="reference", choices=["reference", "split", "f16"],
                   help="reference (default; 'split' is the same mode): the contract mode -- every stage carries its operands as f16 hi / lo pairs "
                        "against the exact f16 weights (the reference's fp32 forward to fp32 summation noise; word times equal the CPU reference's); "
                        "f16: operands rounded to f16 once -- 1.9x faster, word times within one frame for ~98.5 %% of the boundaries")
    p.add_argument("--readers", type=int, default=4, help="reader threads (audio decode + tokenisation ahead of the GPU)")
    return p.parse_args(argv)


if __name__ == "__main__":
    infer_dataset(parse_args())
