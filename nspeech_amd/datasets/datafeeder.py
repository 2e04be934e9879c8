"""Minimal synchronous feeder for train.py (the reference's threaded DataFeeder is SURVEY row F2,
scheduled after the hot path): reads an LJSpeech-layout directory (metadata.csv + wavs/,
corpus/ljspeech.py:4-11), computes both spectrograms with the GPU feature kernels, buckets by
length like datafeeder.py:139-147 (groups of batch_size*batch_group_size sorted by frame count) and
pads exactly like datafeeder.py:189-220 (inputs with 0, targets with 0 up to a multiple of r after
+1 frame of silence)."""
import os
import random

import numpy as np

from ..utils import audio
from ..utils.text import text_to_sequence


def _round_up(x, m):
    r = x % m
    return x if r == 0 else x + m - r


def load_ljspeech_metadata(path):
    items = []
    with open(os.path.join(path, "metadata.csv"), encoding="utf-8") as f:
        for line in f:
            parts = line.strip().split("|")
            if len(parts) >= 3:
                items.append((os.path.join(path, "wavs", "%s.wav" % parts[0]), parts[2]))
            elif len(parts) == 2:
                items.append((os.path.join(path, "wavs", "%s.wav" % parts[0]), parts[1]))
    return items


class DataFeeder(object):
    def __init__(self, hparams, ljspeech=None, seed=0):
        self.hp = hparams
        self.items = load_ljspeech_metadata(ljspeech) if ljspeech else []
        self.cache = {}
        self.rng = random.Random(seed)
        self.cleaners = [x.strip() for x in hparams.cleaners.split(",")]
        self._batches = []
        assert self.items, "no training data found"

    def _example(self, idx):
        if idx not in self.cache:
            wav_path, text = self.items[idx]
            wav = audio.load_wav(wav_path)
            lin, mel = audio.spectrogram_and_mel(wav)
            ids = np.asarray(text_to_sequence(text, self.cleaners), dtype=np.int32)
            self.cache[idx] = (ids, mel.T.astype(np.float32), lin.T.astype(np.float32))
        return self.cache[idx]

    def next_batch(self):
        hp = self.hp
        if not self._batches:
            n, r = hp.batch_size, hp.outputs_per_step
            group = [self._example(self.rng.randrange(len(self.items))) for _ in range(n * hp.batch_group_size)]
            group.sort(key=lambda e: e[1].shape[0])
            self._batches = [group[i:i + n] for i in range(0, len(group), n)]
            self.rng.shuffle(self._batches)
        batch = self._batches.pop()
        r = hp.outputs_per_step
        Ti = max(len(e[0]) for e in batch)
        To = _round_up(max(e[1].shape[0] for e in batch) + 1, r)
        N = len(batch)
        inputs = np.zeros((N, Ti), np.int32)
        lengths = np.zeros((N,), np.int32)
        mel = np.zeros((N, To, hp.num_mels), np.float32)
        lin = np.zeros((N, To, hp.num_freq), np.float32)
        for i, (ids, m, l) in enumerate(batch):
            inputs[i, :len(ids)] = ids
            lengths[i] = len(ids)
            mel[i, :m.shape[0]] = m
            lin[i, :l.shape[0]] = l
        return inputs, lengths, mel, lin
