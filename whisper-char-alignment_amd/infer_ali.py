#!/usr/bin/env python3
"""Forced-alignment driver with the reference's command line (infer_ali.py:151-173), running on the
MI355X engine. Same flags, same result JSON ({args..., precision, recall, f1, r_value}) and the same
`-predictions.pkl` schema (infer_ali.py:118-119,139-148), so eval_ali.py keeps working.

    python infer_ali.py --dataset TIMIT --scp scp/test.wav.scp --model medium --weights /path/medium.pt \
        --aggr topk --topk 10 --aligned_unit_type char --medfilt_width 3 --output_dir results/timit
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 infer_ali.py ...      (one rank per GPU)

Additions: --weights (LOCAL openai-format checkpoint; nothing is fetched by name), --random_init,
--batch_size (utterances per micro-batch through the fused wca_align_batch path), --vocab (local
tiktoken file, needed for --aligned_unit_type subword), --teacher.
--teacher asr runs the reference's flow (greedy whisper.decode pre-pass, infer_ali.py:60-68, then teacher-forcing of its
hypothesis; the encoder runs ONCE per utterance, the reference runs it twice); --teacher text (default) teacher-forces
the dataset transcript and skips the pre-pass.
Utterances with max_frames > 1500 or more than 448 tokens are skipped and their id printed
(infer_ali.py:78-81).
"""
import argparse
import datetime
import json
import os
import sys
import time

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
        return WhisperAMD.from_checkpoint(args.weights, device=device, max_batch=args.batch_size)
    if args.random_init:
        from .synthetic import random_state_dict
        dims = dims_for(args.model)
        return WhisperAMD(dims, device=device, max_batch=args.batch_size).load_state_dict(random_state_dict(dims, seed=0))
    raise SystemExit("no weights: pass --weights /local/path/%s.pt (openai-whisper checkpoint; nothing is downloaded by name) "
                     "or --random_init for a dry run" % args.model)


def infer_dataset(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)
    if rank == 0:
        print(args)
    model = load_model(args, device)
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

    def upload(batch):
        smax = max(len(b["pcm"]) for b in batch)
        pcm = np.zeros((len(batch), smax), dtype=np.float32)
        for j, b in enumerate(batch):
            pcm[j, :len(b["pcm"])] = b["pcm"]
        return torch.from_numpy(pcm).to(device), [len(b["pcm"]) for b in batch]

    def start_asr(batch):
        """--teacher asr, stage 1: enqueue log-mel + encoder + cross-K/V of this micro-batch (no host sync). It runs on the
        engine's first stream while the PREVIOUS micro-batch is decoded and aligned on the second one."""
        pcm_dev, n_samples = upload(batch)
        model.encode_batch(pcm=pcm_dev, n_samples=n_samples)
        return batch

    def flush(batch, encoded=False):
        nonlocal corrects, total_preds, total_gts
        if not batch:
            return
        reuse = False
        if args.teacher == "asr":
            # greedy ASR pre-pass (infer_ali.py:60-61); its encoder output is re-used by the alignment below
            if not encoded:
                start_asr(batch)
            results = decode(model, None, asr_options, encoded_batch=len(batch))
            reuse = True
            pcm_dev = n_samples = None
        else:
            pcm_dev, n_samples = upload(batch)
        if reuse:
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
        n_max = max(len(b["tokens"]) for b in batch)
        toks = np.full((len(batch), n_max), tokenizer.eot, dtype=np.int64)
        for j, b in enumerate(batch):
            toks[j, :len(b["tokens"])] = b["tokens"]
        jump, _ = model.align_batch(None if reuse else pcm_dev, None if reuse else n_samples, torch.from_numpy(toks).to(device),
                                    [len(b["tokens"]) for b in batch], [b["max_frames"] for b in batch], opts)
        for j, b in enumerate(batch):
            if b.get("skip"):
                continue
            words, start_times, end_times = words_from_jump_frames(jump[j], b["text_tokens"], tokenizer, args.aligned_unit_type)
            ends_hat = end_times
            local_times[b["index"]] = (start_times, end_times)
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

    pending, in_flight = [], []
    t0 = time.time()
    for n in mine:
        audio, _mel, duration, texts, starts, ends, fid = dataset[n]
        texts = remove_punctuation(texts)
        max_frames = duration // AUDIO_SAMPLES_PER_TOKEN
        if args.teacher == "text":
            transcription = texts  # the dataset transcript is teacher-forced
            text_tokens = encode(transcription, tokenizer, args.aligned_unit_type)
            tokens = [*tokenizer.sot_sequence, tokenizer.no_timestamps, *text_tokens, tokenizer.eot]
        else:
            text_tokens, tokens = None, []  # filled from the ASR hypothesis when the micro-batch is flushed
        if max_frames > MAX_FRAMES or len(tokens) > MAX_LENGTH or max_frames < 1:
            print(fid)
            continue
        if args.default_whisper_timing:  # per-utterance path (timing.py:116-186), not the fused batch path
            mel = log_mel_spectrogram(pad_or_trim(audio), args.n_mels, model=model)
            if args.teacher == "asr":
                transcription = remove_punctuation(decode(model, mel, asr_options).text)
                text_tokens = encode(transcription, tokenizer, args.aligned_unit_type)
                if len(text_tokens) + len(tokenizer.sot_sequence) + 2 > MAX_LENGTH:
                    print(fid)
                    continue
            words, start_times, end_times, _ws, _ = default_find_alignment(model, tokenizer, text_tokens, mel, int(max_frames),
                                                                         medfilt_width=args.medfilt_width)
            local_times[n] = (np.asarray(start_times, dtype=np.float64), np.asarray(end_times, dtype=np.float64))
            c, _ = eval_n1(ends, end_times, args.tolerance)
            total_gts += len(ends)
            total_preds += len(end_times)
            corrects += c
            continue
        pcm = audio.numpy()[:min(duration, len(audio))]
        pending.append(dict(index=n, pcm=pcm, tokens=tokens, text_tokens=text_tokens, max_frames=int(max_frames), texts=texts,
                            starts=starts, ends=ends, fid=fid))
        if len(pending) == args.batch_size:
            if args.teacher == "asr":
                # two-deep pipeline: encode this micro-batch, then decode + align the previous one beside it
                start_asr(pending)
                flush(in_flight, encoded=True)
                in_flight = pending
            else:
                flush(pending)
            pending = []
    if args.teacher == "asr":
        if pending:
            start_asr(pending)
        flush(in_flight, encoded=True)
        flush(pending, encoded=True)
    else:
        flush(pending)
    elapsed = time.time() - t0

    corrects, total_preds, total_gts = _shard.allreduce_counters(corrects, total_preds, total_gts)
    all_times = _shard.allgather_results(local_times)
    if rank == 0:
        precision, recall, f1, r_value, _ = get_seg_metrics(corrects, corrects, total_preds, total_gts)
        results = dict(precision=precision, recall=recall, f1=f1, r_value=r_value)
        print(results)
        print("aligned %d utterances on %d GPU(s) in %.1f s" % (len(all_times), world, elapsed))
        filename = datetime.datetime.fromtimestamp(time.time()).strftime("%Y-%m-%d-%H:%M:%S")
        os.makedirs(args.output_dir, exist_ok=True)
        with open(os.path.join(args.output_dir, filename + ".json"), "w") as f:
            json.dump({**vars(args), **results}, f)
        if args.save_prediction:
            import joblib
            joblib.dump(all_predictions, os.path.join(args.output_dir, filename + "-predictions.pkl"))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


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
    p.add_argument("--teacher", type=str, default="text", choices=["text", "asr"],
                   help="asr: greedy decode pre-pass gives the teacher text (the reference's behaviour); text: dataset transcript")
    return p.parse_args(argv)


if __name__ == "__main__":
    infer_dataset(parse_args())
