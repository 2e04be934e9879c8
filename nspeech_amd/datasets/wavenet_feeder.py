"""Training pieces for simple_wavenet (datasets/WavenetDataFeeder.py:19-167 of the reference).

Same walk as the reference: the item list is read in order and reshuffled at every wrap-around (:148-167); a waveform
is trimmed of leading / trailing silence (`trim_silence`, process.py:45-54, threshold 0.1), padded with
receptive_field zeros in front and cut into pieces of receptive_field + sample_size samples that overlap by the
receptive field (:104-125); pieces go through a shuffling buffer of `queue_size` entries that hands out random
elements once it holds more than min_dequeue_ratio * queue_size (tf.RandomShuffleQueue, :71-83), `batch_size` at a
time.  The mel / linear "local condition" images the reference attaches to every piece (:127-135) feed lc_channels,
which the shipped wavenet.yaml sets to 0 and this build does not implement: they are not computed."""
import random

import numpy as np

from ..utils import audio
from .datafeeder import load_librispeech_corpus, load_ljspeech_metadata, load_vctk_file_names
from .process import trim_silence  # noqa: F401  (process.py:45-54)


class WavenetFeeder(object):
    def __init__(self, hparams, receptive_field, ljspeech=None, vctk=None, librispeech=None, seed=0, loader=None,
                 silence_threshold=0.1):
        self.hp = hparams
        self.rf = int(receptive_field)
        self.sample_size = int(hparams.sample_size)
        self.silence_threshold = silence_threshold
        self.items = load_ljspeech_metadata(ljspeech) if ljspeech else []
        self.items += load_vctk_file_names(vctk) if vctk else []
        self.items += load_librispeech_corpus(librispeech) if librispeech else []
        assert self.items, "No data found"
        pairs = sorted({(dataset, str(spk)) for _, _, spk, dataset in self.items})
        self.speaker2id = {v: k for k, v in enumerate(pairs)}
        self._rng = random.Random(seed)
        self._offset = 0
        self._loader = loader or audio.load_wav
        self._pool = []           # the shuffling buffer: (piece, speaker id)
        self.capacity = int(hparams.queue_size)
        self.min_after = int(hparams.min_dequeue_ratio * hparams.queue_size)
        self.speaker_ids = None

    def _next_pieces(self):
        if self._offset >= len(self.items):
            self._offset = 0
            self._rng.shuffle(self.items)
        path, _text, spk, dataset = self.items[self._offset]
        self._offset += 1
        wav = np.asarray(self._loader(path), np.float32)
        if self.silence_threshold is not None:
            wav = trim_silence(wav, self.silence_threshold)
        wav = np.pad(wav, [self.rf, 0], "constant")
        sid = self.speaker2id[dataset, str(spk)]
        out = []
        while len(wav) > self.rf + self.sample_size:
            out.append((wav[:self.rf + self.sample_size].copy(), sid))
            wav = wav[self.sample_size:]
        return out

    @property
    def size(self):
        return len(self._pool)

    def next_batch(self):
        """float32 [batch_size, receptive_field + sample_size]; .speaker_ids [batch_size]."""
        n = int(self.hp.batch_size)
        idle = 0
        while len(self._pool) < max(n + self.min_after, 1):
            got = self._next_pieces()
            room = max(self.capacity, n + self.min_after) - len(self._pool)
            self._pool.extend(got[:max(room, 0)] if room < len(got) else got)
            idle = idle + 1 if not got else 0
            if idle > 2 * len(self.items):
                raise RuntimeError("no waveform is longer than sample_size after trimming silence")
        picks = sorted(self._rng.sample(range(len(self._pool)), n), reverse=True)
        batch = [self._pool.pop(i) for i in picks]
        self.speaker_ids = np.asarray([b[1] for b in batch], np.int32)
        return np.stack([b[0] for b in batch])
