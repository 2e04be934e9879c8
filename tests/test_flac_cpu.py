"""ns_flac_decode / audio._load_flac (host code in libnspeech_hip.so; the reference reads LibriSpeech .flac through
librosa.core.load, datasets/corpus/ljspeech.py:17) against streams written by tests/flac_writer.py, an encoder written
from the format specification that lets a test choose every coding decision: all subframe types, fixed orders 0-4,
LPC, both Rice methods, escape partitions, wasted bits, the three stereo decorrelations, odd block sizes, 8 / 16 / 24
bit samples, missing totals / MD5, metadata padding - and against corrupted input (CRC-8, CRC-16, MD5, truncation)."""
import os

import numpy as np
import pytest

import flac_writer as FW


def _speechlike(n, seed, bps=16, channels=1):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = np.zeros((n, channels))
    for c in range(channels):
        f0 = 110 + 30 * c + 20 * np.sin(2 * np.pi * 1.5 * t)
        ph = 2 * np.pi * np.cumsum(f0) / 16000.0
        x[:, c] = sum(np.sin(k * ph) / k for k in range(1, 9)) * (0.3 + 0.2 * np.sin(2 * np.pi * 3 * t)) + 0.02 * rng.standard_normal(n)
    x = x / np.abs(x).max() * 0.8
    return np.round(x * (1 << (bps - 1))).astype(np.int64)


def _lpc(block, order, prec=12):
    x = np.asarray(block, np.float64)
    A = np.stack([x[order - 1 - j:len(x) - 1 - j] for j in range(order)], 1)
    a = np.linalg.lstsq(A, x[order:], rcond=None)[0]
    shift = max(0, min(15, prec - 2 - int(np.ceil(np.log2(np.abs(a).max() + 1e-9)))))
    q = np.clip(np.round(a * (1 << shift)), -(1 << (prec - 1)), (1 << (prec - 1)) - 1).astype(int)
    return dict(type="lpc", order=order, coefs=q.tolist(), shift=shift, prec=prec)


def _decode(tmp_path, blob, name="a.flac"):
    from nspeech_amd.utils import audio as A
    p = os.path.join(str(tmp_path), name)
    with open(p, "wb") as f:
        f.write(blob)
    return A._load_flac(p)


def test_librispeech_like_mono_lpc(tmp_path):
    pcm = _speechlike(4096 * 3 + 1234, 1)
    frames = []
    pos = 0
    for size in (4096, 4096, 4096, 1234):
        spec = _lpc(pcm[pos:pos + size, 0], 8)
        spec.update(porder=3 if size == 4096 else 0)
        frames.append(dict(size=size, subframes=[spec]))
        pos += size
    x, sr = _decode(tmp_path, FW.encode(pcm, 16, 16000, frames))
    assert sr == 16000 and x.shape == (len(pcm), 1)
    assert np.array_equal(np.round(x[:, 0] * 32768).astype(np.int64), pcm[:, 0])


def test_every_subframe_type_and_stereo_mode(tmp_path):
    n = 576 * 2 + 256 * 4 + 192 + 100 + 17
    pcm = _speechlike(n, 2, channels=2)
    pcm[576:1152] = (pcm[576:1152] >> 2) << 2            # two wasted bits in that block
    pcm[1152:1408, 1] = -345                               # a constant right channel
    frames = [
        dict(size=576, assignment="independent", subframes=[dict(type="fixed", order=0, porder=2), dict(type="fixed", order=1)]),
        dict(size=576, assignment="left_side", subframes=[dict(type="fixed", order=2, wasted=2, method=1, porder=1),
                                                         dict(type="verbatim", wasted=2)]),
        dict(size=256, assignment="independent", subframes=[dict(type="fixed", order=3, porder=4), dict(type="constant")]),
        dict(size=256, assignment="side_right", subframes=[dict(type="fixed", order=4, method=1, params=[17]),
                                                          dict(type="fixed", order=2, porder=1, params=[("esc", 17), 9])]),
        dict(size=256, assignment="mid_side", subframes=[_lpc(((pcm[1664:1920, 0] + pcm[1664:1920, 1]) >> 1), 12, prec=15),
                                                        dict(type="fixed", order=1, porder=3)]),
        dict(size=256, assignment="mid_side", bps_from_streaminfo=True,
             subframes=[dict(type="verbatim"), dict(type="fixed", order=0, params=[("esc", 18)])]),
        dict(size=192, assignment="independent", subframes=[_lpc(pcm[2176:2368, 0], 1, prec=5), _lpc(pcm[2176:2368, 1], 32, prec=14)]),
        dict(size=100, assignment="left_side", subframes=[dict(type="fixed", order=4), dict(type="fixed", order=4)]),
        dict(size=17, assignment="independent", subframes=[dict(type="fixed", order=4), dict(type="verbatim")]),
    ]
    for variable in (False, True):
        x, sr = _decode(tmp_path, FW.encode(pcm, 16, 22050, frames, variable=variable, padding_block=40))
        assert sr == 22050 and np.array_equal(np.round(x * 32768).astype(np.int64), pcm), variable


@pytest.mark.parametrize("bps", [8, 24])
def test_other_sample_sizes_and_missing_totals(tmp_path, bps):
    pcm = _speechlike(1024 + 300, 3 + bps, bps=bps, channels=2)
    frames = [dict(size=1024, assignment="mid_side", rate_field=13, subframes=[dict(type="fixed", order=2, porder=2), dict(type="fixed", order=2)]),
              dict(size=300, assignment="independent", rate_field=12, subframes=[_lpc(pcm[1024:, 0], 6), dict(type="verbatim")])]
    x, sr = _decode(tmp_path, FW.encode(pcm, bps, 32000, frames, total_in_header=False, md5=False))
    assert sr == 32000 and np.array_equal(np.round(x * (1 << (bps - 1))).astype(np.int64), pcm)


def test_corrupt_streams_are_refused(tmp_path):
    from nspeech_amd._lib import NSError
    pcm = _speechlike(2048, 5)
    frames = [dict(size=1024, subframes=[dict(type="fixed", order=2, porder=2)]) for _ in range(2)]
    blob = bytearray(FW.encode(pcm, 16, 16000, frames))
    x, _ = _decode(tmp_path, bytes(blob))
    assert x.shape == (2048, 1)
    first = 4 + 4 + 34
    bad = bytearray(blob); bad[first + 4] ^= 0x01                       # the frame number: header CRC-8
    with pytest.raises(NSError, match="CRC-8"):
        _decode(tmp_path, bytes(bad))
    bad = bytearray(blob); bad[first + 40] ^= 0x01                      # a residual bit: frame CRC-16 (or lost sync behind it)
    with pytest.raises(NSError):
        _decode(tmp_path, bytes(bad))
    with pytest.raises((NSError, ValueError)):                         # cut inside the second frame
        _decode(tmp_path, bytes(blob[:len(blob) - 50]))
    bad = bytearray(blob); bad[4 + 4 + 18] ^= 0xFF                      # the stored MD5
    with pytest.raises(ValueError, match="MD5"):
        _decode(tmp_path, bytes(bad))
    with pytest.raises(NSError, match="fLaC"):
        _decode(tmp_path, b"RIFF" + bytes(60))


def test_load_wav_takes_flac_files(tmp_path):
    """audio.load_wav (audio.py:13-14) on a .flac at the model's sample rate: stereo mixed down, offset / duration in
    seconds at the native rate, no resampling needed."""
    from nspeech_amd import hparams
    from nspeech_amd.utils import audio as A
    hp = hparams.load("taco2")
    sr = hp.sample_rate
    pcm = _speechlike(4096, 7, channels=2)
    frames = [dict(size=4096, assignment="mid_side", subframes=[dict(type="fixed", order=2, porder=3), dict(type="fixed", order=1)])]
    p = os.path.join(str(tmp_path), "u.flac")
    with open(p, "wb") as f:
        f.write(FW.encode(pcm, 16, sr, frames))
    x = A.load_wav(p)
    want = (pcm.astype(np.float32) / 32768.0).mean(axis=1)
    assert x.shape == (4096,) and np.allclose(x, want, atol=1e-7)
    y = A.load_wav(p, offset=0.05, duration=0.1)
    s = int(0.05 * sr)
    assert np.allclose(y, want[s:s + int(0.1 * sr)], atol=1e-7)


def test_id3_tags_and_a_missing_total(tmp_path):
    """a leading ID3v2 block and a trailing ID3v1 block are skipped, and a stream whose STREAMINFO leaves the
    total out is decoded into a buffer that grows on demand (a long CONSTANT-coded stream: far more than the first
    guess of max(65536, file size) samples from a few hundred bytes)."""
    n = 4096 * 40
    pcm = np.full((n, 1), 1234, np.int64)
    frames = [dict(size=4096, subframes=[dict(type="constant")]) for _ in range(40)]
    blob = FW.encode(pcm, 16, 16000, frames, total_in_header=False)
    assert len(blob) < 2000
    tag2 = b"ID3" + bytes([3, 0, 0]) + bytes([0, 0, 1, 5]) + bytes(133)            # sync-safe size 1*128 + 5 = 133
    tag1 = b"TAG" + bytes(125)
    for name, data in (("plain.flac", blob), ("tagged.flac", tag2 + blob + tag1)):
        x, sr = _decode(tmp_path, data, name)
        assert sr == 16000 and x.shape == (n, 1) and np.all(np.round(x * 32768) == 1234), name
