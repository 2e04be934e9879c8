"""Host logic of the WaveNet training pipeline (datasets/WavenetDataFeeder.py:104-125, process.py:45-54): silence
trimming, the receptive-field padded pieces and the shuffling buffer.  No GPU: waveforms come from a stub loader."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_trim_silence_cuts_at_frame_boundaries():
    from nspeech_amd.datasets.wavenet_feeder import trim_silence
    wav = np.zeros(20000, np.float32)
    wav[6000:12000] = 0.5 * np.sin(np.arange(6000) * 0.3)
    out = trim_silence(wav, 0.1)
    # rms of a centred 2048-frame: frames k*512 whose window [k*512 - 1024, k*512 + 1024) holds enough of the tone
    energy = [np.sqrt(np.mean(np.pad(wav, 1024, mode="reflect")[k * 512:k * 512 + 2048] ** 2)) for k in range(1 + 20000 // 512)]
    loud = [k for k, e in enumerate(energy) if e > 0.1]
    assert len(out) == (loud[-1] - loud[0]) * 512 and np.array_equal(out, wav[loud[0] * 512:loud[-1] * 512])
    assert trim_silence(np.zeros(5000, np.float32), 0.1).size == 0
    short = 0.5 * np.ones(100, np.float32)                 # shorter than a frame: one frame of its own length
    assert trim_silence(short, 0.1).size == 0              # a single loud frame gives wav[i:i]


def test_pieces_overlap_by_the_receptive_field(tmp_path):
    from nspeech_amd import hparams as H
    from nspeech_amd.datasets.wavenet_feeder import WavenetFeeder
    hp = H.load("wavenet")
    hp.sample_size, hp.batch_size, hp.queue_size = 100, 4, 16
    os.makedirs(tmp_path / "wavs")
    with open(tmp_path / "metadata.csv", "w") as f:
        f.write("A|x|x\nB|y|y\n")
    waves = {"A.wav": (0.5 * np.sin(np.arange(1, 1001) * 0.37)).astype(np.float32),
             "B.wav": (0.5 * np.cos(np.arange(1, 701) * 0.21)).astype(np.float32)}
    rf = 50
    fd = WavenetFeeder(hp, rf, ljspeech=str(tmp_path), loader=lambda p: waves[os.path.basename(p)], silence_threshold=None)
    pieces = fd._next_pieces()
    # 1000 samples behind 50 zeros, pieces of 150 every 100 while MORE than 150 remain: 9 pieces
    padded = np.concatenate([np.zeros(rf, np.float32), waves["A.wav"]])
    assert len(pieces) == 9
    for i, (p, sid) in enumerate(pieces):
        assert sid == 0 and np.array_equal(p, padded[i * 100:i * 100 + 150])
    batch = fd.next_batch()
    assert batch.shape == (4, 150) and batch.dtype == np.float32 and fd.speaker_ids.shape == (4,)
    padded_b = np.concatenate([np.zeros(rf, np.float32), waves["B.wav"]])
    valid = [padded[i * 100:i * 100 + 150] for i in range(9)] + [padded_b[i * 100:i * 100 + 150] for i in range(6)]
    assert all(any(np.array_equal(b, v) for v in valid) for b in batch)          # every row is a whole piece
    assert fd.size <= max(fd.capacity, 4 + fd.min_after)
