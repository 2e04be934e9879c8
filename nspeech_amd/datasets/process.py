"""Utterance processing in front of the feeder (datasets/process.py:23-68 of the reference): load -> trim the silent ends
-> both spectrograms.  `trim_wav` is what the training path uses (process.py:27); `trim_silence` (process.py:45-54) is the
WaveNet feeder's.

Frame energies come from one running sum of squares (float64) instead of a frame matrix: a 10 s utterance is 391 frames of
1024 samples, and the decision per frame is a comparison of two sums, so the intervals are integers that either agree with
the oracle's per-frame loops or do not (`tests/test_feeder_cpu.py`)."""
import os

import numpy as np

from ..utils import audio


def _loud_frames(wav, top_db, frame_length, hop_length):
    """[3P] librosa 0.6.0 effects._signal_to_frame_nonsilent with ref = np.max: frames (centred, reflect padded) whose mean
    square is within top_db of the loudest frame's; both sides of the ratio are floored at 1e-10 (power_to_db's amin)."""
    y = np.pad(np.asarray(wav, np.float64), frame_length // 2, mode="reflect")
    n_frames = 1 + (len(y) - frame_length) // hop_length
    sq = np.concatenate([[0.0], np.cumsum(y * y)])
    starts = np.arange(n_frames) * hop_length
    mse = np.maximum((sq[starts + frame_length] - sq[starts]) / frame_length, 0.0)
    amin = 1e-10
    return 10.0 * np.log10(np.maximum(amin, mse)) - 10.0 * np.log10(max(amin, float(mse.max()))) > -top_db


def split(wav, top_db=60, frame_length=2048, hop_length=512):
    """[3P] librosa 0.6.0 effects.split: [start, end) sample intervals of the runs of loud frames, clipped to the signal."""
    loud = _loud_frames(wav, top_db, frame_length, hop_length)
    edges = np.flatnonzero(np.diff(loud.astype(np.int64))) + 1
    edges = np.concatenate([[0] if loud[0] else [], edges, [len(loud)] if loud[-1] else []]).astype(np.int64)
    return np.minimum(edges * hop_length, len(wav)).reshape(-1, 2)


def trim_silence(wav, threshold, frame_length=2048, hop_length=512):
    """process.py:45-54 over librosa.feature.rmse [3P, librosa 0.6: centred frames of 2048 every 512 samples, reflect
    padded] -> wav[first loud frame * 512 : last loud frame * 512]; all silence -> empty."""
    if wav.size < frame_length:
        frame_length = wav.size
    if wav.size == 0:
        return wav
    y = np.pad(wav.astype(np.float64), frame_length // 2, mode="reflect")
    n_frames = 1 + (len(y) - frame_length) // hop_length
    sq = np.concatenate([[0.0], np.cumsum(y * y)])
    starts = np.arange(n_frames) * hop_length
    energy = np.sqrt((sq[starts + frame_length] - sq[starts]) / frame_length)
    loud = np.nonzero(energy > threshold)[0] * hop_length
    return wav[loud[0]:loud[-1]] if loud.size else wav[:0]


def _find_start(splits, min_samples=2000):
    """process.py:56-60."""
    for split_start, split_end in splits:
        if split_end - split_start > min_samples:
            return max(0, int(split_start) - min_samples)
    return 0


def _find_end(splits, num_samples, min_samples=2000):
    """process.py:63-67."""
    for split_start, split_end in reversed(list(splits)):
        if split_end - split_start > min_samples:
            return min(num_samples, int(split_end) + min_samples)
    return num_samples


def trim_wav(wav, threshold_db=25):
    """process.py:39-42: trims silence from the ends of the wav (the second positional argument of librosa's split is
    top_db; frames of 1024 every 512 samples)."""
    splits = split(wav, threshold_db, frame_length=1024, hop_length=512)
    return wav[_find_start(splits):_find_end(splits, len(wav))]


def process_utterance(wav_path, dataset_id=None, loader=None):
    """process.py:23-36: (id, trimmed wav, linear [T, F], mel [T, M], n_frames); both spectrograms from ONE pass of the
    fused GPU feature kernel."""
    idx = os.path.basename(wav_path)[:-4]
    wav = trim_wav((loader or audio.load_wav)(wav_path))
    lin, mel = audio.spectrogram_and_mel(wav)
    return idx, wav, lin.T, mel.T, lin.shape[1]


def build_from_path(filenames, num_workers=1, tqdm=lambda x: x, limit=0):
    """process.py:10-18.  The reference farms utterances out to worker processes because its two librosa STFTs are host
    work; here the features are one GPU launch per utterance, so the walk is sequential (num_workers is accepted and
    ignored).  Keeps the reference's `len(futures) > limit` cut, i.e. limit + 1 items."""
    out = []
    for wav_path, _text, _speaker, dataset_id in filenames:
        if limit and len(out) > limit:
            break
        out.append(process_utterance(wav_path, dataset_id))
    return list(tqdm(out))
