"""Training data feeder (datasets/datafeeder.py:19-220 of the reference).

Same batching semantics as the reference:
  * the item list is walked in order and reshuffled at every wrap-around (`_get_next_example`,
    datafeeder.py:157-181); processed utterances are cached in RAM (`processed_data`);
  * a group of batch_size * batch_group_size examples is sorted by target length and cut into
    batches, the batches are shuffled, and every batch is shuffled internally
    (`_enqueue_next_group` :139-147, `_prepare_batch` :190);
  * inputs are padded with 0 to the longest, targets with 0 to (longest + 1) rounded up to a
    multiple of outputs_per_step (`_prepare_targets` :204-206, `_round_up` :218-220);
  * with a CMUDict (`use_cmudict`, commented out in the reference, :96-108) half of the sentences
    get each word replaced by its `{ARPAbet}` spelling with probability 0.5 (:178-186);
  * a background thread keeps up to 8 prepared batches queued (the reference's FIFOQueue(8), :72).

MI355X-side differences: both spectrograms of an utterance come from ONE pass of the fused GPU
feature kernel (`audio.spectrogram_and_mel`) instead of two librosa STFTs in worker threads, and for
data-parallel runs the sorted group is dealt round-robin over the ranks so that every
rank sees the same length mix: rank r takes examples r, r + world, ... of the sorted group."""
import glob
import os
import queue
import random
import re
import threading

import numpy as np

from ..utils import audio
from ..utils.text import text_to_sequence
from .process import trim_wav

_p_cmudict = 0.5      # datafeeder.py:16
_pad = 0              # datafeeder.py:17


def _round_up(x, m):
    r = x % m
    return x if r == 0 else x + m - r


def load_ljspeech_metadata(path):
    """corpus/ljspeech.py:4-11: `id|raw text|normalised text` -> (wavs/id.wav, normalised text, speaker 0,
    "ljspeech")."""
    items = []
    with open(os.path.join(path, "metadata.csv"), encoding="utf-8") as f:
        for line in f:
            parts = line.strip().split("|")
            if len(parts) >= 3:
                items.append((os.path.join(path, "wavs", "%s.wav" % parts[0]), parts[2], 0, "ljspeech"))
            elif len(parts) == 2:
                items.append((os.path.join(path, "wavs", "%s.wav" % parts[0]), parts[1], 0, "ljspeech"))
    return items


def load_vctk_file_names(path):
    """corpus/vctk.py:11-20: wav48/pNNN/pNNN_MMM.wav with its transcript under txt/; speaker = NNN."""
    items = []
    for wav_path in sorted(glob.glob("%s/wav48/p*/*.wav" % path)):
        text_path = wav_path.replace("wav48", "txt").replace("wav", "txt")
        if os.path.isfile(text_path):
            with open(text_path, "r") as f:
                text = f.read().strip()
            name = os.path.splitext(os.path.basename(wav_path))[0]
            items.append((wav_path, text, re.match(r"p([0-9]+)_", name).group(1), "vctk"))
    return items


def load_librispeech_corpus(path):
    """corpus/ljspeech.py:14-27 (`load_libre_2`): corpus.csv rows `speaker-chapter-utterance,path,text,mode`."""
    items = []
    with open(os.path.join(path, "corpus.csv"), encoding="utf-8") as f:
        for line in f:
            identifier, rel, text, _mode = line.strip().split(",")
            items.append((os.path.join(path, rel), text, identifier.split("-")[0], "libre"))
    return items


def _pinned_zeros(shape):
    """Zero-filled float32 array in page-locked host memory (torch's caching host allocator; the array keeps the tensor
    alive), so that the upload of a prepared batch is one asynchronous DMA with no staging copy."""
    import torch
    return torch.zeros(shape, dtype=torch.float32, pin_memory=True).numpy()


def prepare_batch(batch, outputs_per_step, rng, longest=0, alloc=None):
    """datafeeder.py:189-220.  batch: list of (ids, speaker_id, mel [T,M], linear [T,F]).
    `longest`: data-parallel runs pass the longest target of the GLOBAL batch (all ranks), so that every rank pads to
    the same T_out and the mean of the ranks' unmasked mean losses is the global mean."""
    batch = list(batch)
    rng.shuffle(batch)
    Ti = max(len(e[0]) for e in batch)
    To = _round_up(max(max(e[3].shape[0] for e in batch), longest) + 1, outputs_per_step)
    N = len(batch)
    inputs = np.full((N, Ti), _pad, np.int32)
    lengths = np.zeros((N,), np.int32)
    speakers = np.zeros((N,), np.int32)
    if alloc is None:
        mel = np.full((N, To, batch[0][2].shape[1]), _pad, np.float32)
        lin = np.full((N, To, batch[0][3].shape[1]), _pad, np.float32)
    else:                       # _pad is 0
        mel, lin = alloc((N, To, batch[0][2].shape[1])), alloc((N, To, batch[0][3].shape[1]))
    for i, (ids, spk, m, l) in enumerate(batch):
        inputs[i, :len(ids)] = ids
        lengths[i] = len(ids)
        speakers[i] = spk
        mel[i, :m.shape[0]] = m
        lin[i, :l.shape[0]] = l
    return inputs, lengths, speakers, mel, lin


def prepare_batch_device(batch, outputs_per_step, rng, longest, device, stream):
    """prepare_batch for examples whose features are CUDA tensors (the HBM-resident cache): the padded target arrays are
    assembled ON the device - a zero fill and one device-to-device copy per utterance, issued on the feeder's own
    stream - so a training batch never crosses PCIe.  Returns the batch with device targets and the event that marks
    them complete."""
    import torch
    batch = list(batch)
    rng.shuffle(batch)
    Ti = max(len(e[0]) for e in batch)
    To = _round_up(max(max(int(e[3].shape[0]) for e in batch), longest) + 1, outputs_per_step)
    N = len(batch)
    inputs = np.full((N, Ti), _pad, np.int32)
    lengths = np.zeros((N,), np.int32)
    speakers = np.zeros((N,), np.int32)
    with torch.cuda.stream(stream):
        mel = torch.zeros((N, To, batch[0][2].shape[1]), dtype=torch.float32, device=device)
        lin = torch.zeros((N, To, batch[0][3].shape[1]), dtype=torch.float32, device=device)
        for i, (ids, spk, m, l) in enumerate(batch):
            inputs[i, :len(ids)] = ids
            lengths[i] = len(ids)
            speakers[i] = spk
            mel[i, :m.shape[0]].copy_(m)
            lin[i, :l.shape[0]].copy_(l)
        ev = torch.cuda.Event()
        ev.record(stream)
    return inputs, lengths, speakers, mel, lin, ev


class DataFeeder(object):
    """next_batch() -> (inputs [N,T_in] int32, input_lengths [N], mel [N,T_out,M], linear [N,T_out,F]);
    `speaker_ids` of the last batch are kept in .speaker_ids (single-speaker corpora: zeros)."""

    def __init__(self, hparams, ljspeech=None, seed=0, rank=0, world=1, cmudict=None, prefetch=True, features=None,
                 loader=None, vctk=None, librispeech=None, device=None, pinned=False, device_cache=False, trim=True):
        self.hp = hparams
        # device_cache: the features of every utterance stay in HBM where the feature kernel wrote them (the reference
        # keeps them in host RAM, `processed_data`, datafeeder.py:165-176) and batches are assembled on the device.
        # All of LJSpeech is 24 h x 80 frames/s x 1105 floats = 30.5 GB of float32 - a tenth of one MI355X's 288 GB.
        self._device_cache = bool(device_cache) and features is None
        self._stream = None
        self._alloc = _pinned_zeros if pinned else None      # targets prepared in page-locked memory (DeviceStager)
        # the GPU the feature kernels of the prefetch thread must run on: torch's current device is per host thread,
        # so the worker binds it itself (a rank that called set_device(local) only in its main thread would otherwise
        # extract features on GPU 0)
        if device is None:
            try:
                import torch
                device = torch.cuda.current_device() if torch.cuda.is_available() else None
            except Exception:
                device = None
        self.device = device
        # datafeeder.py:44-53: every corpus named on the command line contributes its items
        self.items = load_ljspeech_metadata(ljspeech) if ljspeech else []
        self.items += load_vctk_file_names(vctk) if vctk else []
        self.items += load_librispeech_corpus(librispeech) if librispeech else []
        assert self.items, "no training data found"
        # datafeeder.py:58-61 numbers the (corpus, speaker) pairs in set order and pins that order in a joblib cache;
        # here the pairs are numbered in sorted order, which every data-parallel rank derives identically
        pairs = sorted({(dataset, str(spk)) for _, _, spk, dataset in self.items})
        self.id2speaker = dict(enumerate(pairs))
        self.speaker2id = {v: k for k, v in self.id2speaker.items()}
        self.rank, self.world = rank, world
        self.cleaners = [x.strip() for x in hparams.cleaners.split(",")]
        self.cache = {}                       # wav path -> (mel, linear), the reference's processed_data
        # every rank walks the SAME shuffled item order (shared seed) and deals the sorted group; the
        # CMUDict coin flips and in-batch shuffles use a rank-local generator
        self._order_rng = random.Random(seed)
        self._rng = random.Random(seed * 7919 + rank)
        self._offset = 0
        self._cmudict = cmudict
        self._features = features or audio.spectrogram_and_mel
        self._loader = loader or audio.load_wav
        self._trim = trim         # False: the corpus is already trimmed (the reference always trims, process.py:27)
        self.speaker_ids = None
        self._queue = queue.Queue(maxsize=8) if prefetch else None
        self._stop = False
        self._thread = None
        self._pending = []
        self._error = None

    # ------------------------------------------------------------------ examples
    def _maybe_get_arpabet(self, word):
        arpabet = self._cmudict.lookup(word)
        return "{%s}" % arpabet[0] if arpabet is not None and self._rng.random() < 0.5 else word

    def _get_next_example(self):
        if self._offset >= len(self.items):
            self._offset = 0
            self._order_rng.shuffle(self.items)
        wav_path, text, local_speaker, dataset = self.items[self._offset]
        self._offset += 1
        if wav_path not in self.cache:
            if self._device_cache:
                import torch
                with torch.cuda.stream(self._feeder_stream()):
                    lin, mel = audio.spectrogram_and_mel_device(self._wav(wav_path))      # [T, F], [T, M] in HBM
                self.cache[wav_path] = (mel, lin)
            else:
                lin, mel = self._features(self._wav(wav_path))
                self.cache[wav_path] = (np.ascontiguousarray(mel.T, np.float32), np.ascontiguousarray(lin.T, np.float32))
        mel, lin = self.cache[wav_path]
        if self._cmudict and self._rng.random() < _p_cmudict:
            text = " ".join(self._maybe_get_arpabet(w) for w in text.split(" "))
        ids = np.asarray(text_to_sequence(text, self.cleaners), dtype=np.int32)
        return ids, self.speaker2id[dataset, str(local_speaker)], mel, lin

    def _wav(self, wav_path):
        """process.py:27: the silent ends are cut before the features are taken."""
        wav = self._loader(wav_path)
        return trim_wav(wav) if self._trim else wav

    def _feeder_stream(self):
        if self._stream is None:
            import torch
            self._stream = torch.cuda.Stream(device=self.device)
        return self._stream

    def _prepare(self, batch, r, longest=0):
        if self._device_cache:
            import torch
            return prepare_batch_device(batch, r, self._rng, longest, torch.device("cuda", self.device), self._feeder_stream())
        return prepare_batch(batch, r, self._rng, longest, alloc=self._alloc)

    def _next_group(self):
        n, r = self.hp.batch_size, self.hp.outputs_per_step
        per_rank = n * self.hp.batch_group_size
        if self.world == 1:
            examples = [self._get_next_example() for _ in range(per_rank)]
            examples.sort(key=lambda e: e[3].shape[0])
        else:
            # every rank walks the same per_rank * world items (cheap for the ones it does not keep: the target
            # length is all the sort needs, but the features are cached anyway), sorts, and keeps its share
            group = [self._get_next_example() for _ in range(per_rank * self.world)]
            group.sort(key=lambda e: e[3].shape[0])
            examples = group[self.rank::self.world]
        batches = [examples[i:i + n] for i in range(0, len(examples), n)]
        if self.world == 1:
            # the batch order comes from the same generator and the same call as in the multi-rank walk below (one
            # shuffle of the batch indices per group); the ITEMS of a group still differ between world sizes, since a
            # group spans per_rank * world items
            order = list(range(len(batches)))
            self._order_rng.shuffle(order)
            return [self._prepare(batches[j], r) for j in order]
        # global batch j = group[j*n*world : (j+1)*n*world]; every rank shuffles the batch order with the SHARED
        # generator (global step k is the same batch j on all ranks) and pads to the longest target of the whole
        # global batch, so T_out is equal across ranks and averaging the rank gradients is the global mean loss
        longest = [max(e[3].shape[0] for e in group[j * n * self.world:(j + 1) * n * self.world])
                   for j in range(len(batches))]
        order = list(range(len(batches)))
        self._order_rng.shuffle(order)
        return [self._prepare(batches[j], r, longest[j]) for j in order]

    # ------------------------------------------------------------------ queue
    def _worker(self):
        try:
            if self.device is not None:
                import torch
                torch.cuda.set_device(self.device)
            while not self._stop:
                for b in self._next_group():
                    while not self._stop:
                        try:
                            self._queue.put(b, timeout=0.1)
                            break
                        except queue.Full:
                            pass
                    if self._stop:
                        break
        except Exception as e:              # surfaced by next_batch()
            self._error = e
            self._queue.put(None)

    def stop(self):
        """End the background thread (the reference's coord.request_stop, datafeeder.py:96-101) and drop what it had queued:
        a feeder that is merely abandoned keeps its thread, its queued batches and their pinned / device memory alive."""
        self._stop = True
        t, self._thread = self._thread, None
        if t is not None:
            while t.is_alive():
                try:
                    self._queue.get_nowait()
                except queue.Empty:
                    pass
                t.join(timeout=0.05)
        if self._queue is not None:
            while not self._queue.empty():
                self._queue.get_nowait()

    def start(self):
        if self._queue is not None and self._thread is None:
            self._stop = False
            self._thread = threading.Thread(target=self._worker, name="datafeeder", daemon=True)
            self._thread.start()
        return self

    def next_batch(self):
        if self._queue is not None:
            self.start()
            b = self._queue.get()
            if b is None:
                raise self._error
        else:
            if not self._pending:
                self._pending = self._next_group()
            b = self._pending.pop(0)
        if len(b) == 6:           # assembled on the device: the caller's stream waits for the feeder's copies
            import torch
            inputs, lengths, speakers, mel, lin, ev = b
            main = torch.cuda.current_stream(mel.device)
            main.wait_event(ev)
            mel.record_stream(main)
            lin.record_stream(main)
        else:
            inputs, lengths, speakers, mel, lin = b
        self.speaker_ids = speakers
        return inputs, lengths, mel, lin


class DeviceStager(object):
    """Targets of batch k+1 travel to the GPU while step k computes: the feeder's NumPy batch is copied into pinned
    host memory (by the feeder's own thread when it was built with pinned=True: prepare_batch then fills page-locked
    arrays directly) and uploaded with asynchronous copies on a COPY stream of its own; next_batch() makes the compute
    stream wait for that copy's event and returns device tensors for the two big arrays (mel + linear targets, 136 MB
    per step at the LJSpeech batch shape - 2.2 ms of PCIe time that would otherwise sit in front of every step) and the
    small ones (ids, lengths, speakers) as they came.  The reference's counterpart is the FIFOQueue + feed_dict of
    datafeeder.py:72-88, which TensorFlow drains into device memory on its own threads."""

    def __init__(self, feeder, device):
        import torch
        self.feeder, self.torch = feeder, torch
        self.device = torch.device(device)
        from .. import ops
        self.copy_stream = ops.concurrent_stream(self.device)      # one that really runs beside the compute stream
        self.speaker_ids = None
        self._next = None

    def _stage(self):
        torch = self.torch
        inputs, lengths, mel, lin = self.feeder.next_batch()
        speakers = self.feeder.speaker_ids
        if torch.is_tensor(mel):        # HBM-resident feature cache: the batch was assembled on the device
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            return inputs, lengths, speakers, mel, lin, ev, None
        pm, pl = torch.from_numpy(mel), torch.from_numpy(lin)
        if not pm.is_pinned():          # a feeder built without pinned=True: one staging copy here, in the caller's thread
            pm, pl = pm.pin_memory(), pl.pin_memory()
        with torch.cuda.stream(self.copy_stream):
            dm = pm.to(self.device, non_blocking=True)
            dl = pl.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return inputs, lengths, speakers, dm, dl, ev, (pm, pl)

    def next_batch(self):
        """(inputs, lengths, mel [device], linear [device]); .speaker_ids as DataFeeder.  Stages the batch after it."""
        torch = self.torch
        cur = self._next or self._stage()
        inputs, lengths, speakers, dm, dl, ev, _pinned = cur
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ev)
        dm.record_stream(main)          # allocated on the copy stream, consumed on the compute stream
        dl.record_stream(main)
        self.speaker_ids = speakers
        self._next = self._stage()
        return inputs, lengths, dm, dl
