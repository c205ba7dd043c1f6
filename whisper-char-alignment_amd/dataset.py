"""Corpus wrappers with the reference's names (dataset.py:14-122): TIMIT and LibriSpeech datasets driven
by a Kaldi-style `.scp` list (`<fid> <path>` per line, README.md:55-62), plus Collate and an AMI wrapper for the
`ami_kaldi.pkl` references the reference's README describes.

Differences from the reference, all on the I/O side (the per-item tuple is unchanged):
  * audio is read lazily (the reference's TIMIT loads the whole corpus in __init__, dataset.py:26-36);
  * no torchaudio: NIST SPHERE / RIFF WAV are parsed by audio.load_audio (TIMIT '.wav' files are SPHERE) and FLAC
    (LibriSpeech) by the in-tree host decoder csrc/flac.cpp (wca_flac_decode);
  * `ls_alignment_{split}.txt` lines are parsed with ast.literal_eval instead of eval (dataset.py:87);
  * the log-mel is computed on the GPU by the engine (csrc/logmel.hip) instead of torch.stft on the host;
    `__getitem__` returns mel=None when constructed with compute_mel=False (the fused batch path takes PCM).
Item tuple: (audio, mel, duration, text, starts, ends, fid)  -- duration = number of samples before padding.
"""
import ast
import os
from glob import glob

import numpy as np
import torch

from . import audio as _audio


def _load_mono(path):
    pcm, sr = _audio.load_audio(path)  # NIST SPHERE / RIFF WAVE / FLAC, all decoded in-tree
    pcm = np.asarray(pcm, dtype=np.float32)
    if pcm.ndim > 1:
        pcm = pcm.reshape(-1) if pcm.shape[0] == 1 else pcm[0]
    return pcm, sr


class Collate:
    """batch_size=1 collate of the reference (dataset.py:14-18): unwraps the single item."""

    def __call__(self, batch):
        audio, mel, duration, text, starts, ends, fid = list(zip(*batch))
        return audio[0], mel[0], duration[0], text[0], starts[0], ends[0], fid[0]


class _ScpDataset(torch.utils.data.Dataset):
    sample_rate = 16000

    def __init__(self, n_mels=80, device="cpu", model=None, compute_mel=True):
        self.n_mels = n_mels
        self.device = device
        self.model = model
        self.compute_mel = compute_mel
        self.items = []

    def __len__(self):
        return len(self.items)

    def duration_hint(self, i):
        """Cheap length estimate (bytes on disk) used to balance shards; exact duration needs a decode."""
        try:
            return os.path.getsize(self.items[i][0])
        except OSError:
            return 0

    def read(self, i):
        """(pcm f32 numpy [duration] (un-padded, trimmed to 30 s), duration, text, starts, ends, fid): what the fused
        batch path needs, without the 30 s zero padding / log-mel of __getitem__. Thread-safe (reader pool)."""
        path = self.items[i][0]
        pcm, sr = _load_mono(path)
        assert sr == self.sample_rate
        duration = len(pcm)
        text, starts, ends, fid = self._labels(i)
        return pcm[:_audio.N_SAMPLES], duration, text, starts, ends, fid

    def _audio_and_mel(self, path):
        pcm, sr = _load_mono(path)
        assert sr == self.sample_rate
        duration = len(pcm)
        audio = _audio.pad_or_trim(torch.from_numpy(pcm))
        mel = None
        if self.compute_mel:
            mel = _audio.log_mel_spectrogram(audio, self.n_mels, model=self.model)
        return audio, mel, duration


class TIMIT(_ScpDataset):
    """scp lines `<fid> <path/to/x.wav>`; word ground truth in the sibling `.wrd` file
    (`<start_sample> <end_sample> <word>` per line, dataset.py:53-64)."""

    def __init__(self, scp_file="scp/test.wav.scp", n_mels=80, device="cpu", model=None, compute_mel=True):
        super().__init__(n_mels, device, model, compute_mel)
        with open(scp_file) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 2:
                    fid, path = parts[0], parts[1]
                    self.items.append((path, path.split(".wav")[0] + ".wrd", fid))

    def process_text(self, filename):
        starts, ends, words = [], [], []
        with open(filename) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 3:
                    starts.append(float(parts[0]) / self.sample_rate)
                    ends.append(float(parts[1]) / self.sample_rate)
                    words.append(parts[2])
        return " ".join(words), starts, ends

    def _labels(self, i):
        _path, wrd, fid = self.items[i]
        text, starts, ends = self.process_text(wrd)
        return text, starts, ends, fid

    def __getitem__(self, i):
        text, starts, ends, fid = self._labels(i)
        audio, mel, duration = self._audio_and_mel(self.items[i][0])
        return audio, mel, duration, text, starts, ends, fid


class LibriSpeech(_ScpDataset):
    """scp lines `<fid> <root>/<split>/<spk>/<chap>/<fid>.flac`; transcripts from `*.trans.txt`, word
    alignments from `ls_alignment_{split}.txt` lines `<fid> [(word, start, end), ...]` (dataset.py:67-122).
    Audio longer than 30 s is trimmed, as in the reference."""

    def __init__(self, scp_file="scp/dev-clean.wav.scp", n_mels=80, device="cpu", model=None, compute_mel=True, alignment_file=None):
        super().__init__(n_mels, device, model, compute_mel)
        with open(scp_file) as f:
            scp = [l for l in f if l.strip()]
        first_path = scp[0].split()[1]
        split = first_path.split("/")[-4]
        root = first_path.split(split)[0]
        labels = {}
        for trans in sorted(glob(os.path.join(root, split, "**/*.trans.txt"), recursive=True)):
            with open(trans) as f:
                for line in f:
                    fid, text = line.split(" ", 1)
                    labels[fid] = text
        ali = {}
        with open(alignment_file or "ls_alignment_%s.txt" % split) as f:
            for line in f:
                fid, rest = line.split(" ", 1)
                ali[fid] = ast.literal_eval(rest.strip())
        for line in scp:
            fid, path = line.split()[:2]
            self.items.append((path, labels[fid], ali[fid], fid))

    def _labels(self, i):
        _path, _text, ali, fid = self.items[i]
        starts, ends, words = [], [], []
        for item in ali:
            if item[0] == "":
                continue
            words.append(item[0])
            starts.append(item[1])
            ends.append(item[2])
        return " ".join(words), starts, ends, fid

    def __getitem__(self, i):
        text, starts, ends, fid = self._labels(i)
        audio, mel, duration = self._audio_and_mel(self.items[i][0])
        return audio, mel, duration, text, starts, ends, fid


class AMI(_ScpDataset):
    """AMI long-form segments (BASELINE config 4). The reference publishes the Kaldi word alignments as `ami_kaldi.pkl`
    (README.md:64-71): `{"AMI_TS3003d_H03_MTD012ME_0255148_0255515": [("w1", s1, e1), ("w2", s2, e2), ...], ...}` but this
    branch of the reference has no dataset class for it (infer_ali.py:28 lists TIMIT and LibriSpeech only); this one follows
    the LibriSpeech wrapper: scp lines `<segment id> <path/to/segment.wav>`, words / start / end seconds from the pickle
    (empty words dropped), audio beyond 30 s trimmed."""

    def __init__(self, scp_file="scp/ami.wav.scp", n_mels=80, device="cpu", model=None, compute_mel=True, alignment_file="ami_kaldi.pkl"):
        super().__init__(n_mels, device, model, compute_mel)
        import pickle
        with open(alignment_file, "rb") as f:
            ali = pickle.load(f)
        with open(scp_file) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 2:
                    if parts[0] not in ali:
                        raise KeyError("%s has no entry in %s" % (parts[0], alignment_file))
                    self.items.append((parts[1], ali[parts[0]], parts[0]))

    def _labels(self, i):
        _path, ali, fid = self.items[i]
        words = [w for w, _, _ in ali if w != ""]
        starts = [float(s) for w, s, _ in ali if w != ""]
        ends = [float(e) for w, _, e in ali if w != ""]
        return " ".join(words), starts, ends, fid

    def __getitem__(self, i):
        text, starts, ends, fid = self._labels(i)
        audio, mel, duration = self._audio_and_mel(self.items[i][0])
        return audio, mel, duration, text, starts, ends, fid
