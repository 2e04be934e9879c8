"""GPU parity of the audio kernels (STFT/mel features, Griffin-Lim, pre-emphasis) against the
float64 NumPy oracle."""
import numpy as np
import pytest
import torch

from oracle import audio_oracle as AO
from test_audio_oracle import HP, _speechlike

pytestmark = pytest.mark.gpu


@pytest.fixture()
def audio(dev):
    from nspeech_amd import hparams
    from nspeech_amd.utils import audio as A
    hp = hparams.load("taco2")
    return A, hp


def test_features_match_oracle(audio):
    A, hp = audio
    for L, seed in ((20000, 0), (12345, 1), (2600, 2)):
        y = _speechlike(L, seed)
        lin = A.spectrogram(y)
        mel = A.melspectrogram(y)
        rl, rm = AO.spectrogram(y, HP), AO.melspectrogram(y, HP)
        assert lin.shape == rl.shape and mel.shape == rm.shape
        # tolerance: fp32 FFT + log10 on the GPU vs float64; values live in [0, 1]
        assert np.abs(lin - rl).max() < 2e-4, np.abs(lin - rl).max()
        assert np.abs(mel - rm).max() < 2e-4, np.abs(mel - rm).max()
        l2, m2 = A.spectrogram_and_mel(y)
        assert np.array_equal(l2, lin) and np.array_equal(m2, mel)


def test_min_level_db_both_signs(audio):
    A, hp = audio
    y = _speechlike(6000, 4)
    hp.min_level_db = -100
    try:
        got = A.spectrogram(y)
    finally:
        hp.min_level_db = 100
    assert np.abs(got - AO.spectrogram(y, dict(HP, min_level_db=-100))).max() < 2e-4


def test_preemphasis_roundtrip(audio):
    A, hp = audio
    y = _speechlike(50001, 5)
    p = A.preemphasis(y)
    assert np.abs(p - AO.preemphasis(y, 0.97)).max() < 1e-6
    q = A.inv_preemphasis(y)
    ref = AO.inv_preemphasis(y, 0.97)
    assert np.abs(q - ref).max() < 2e-4 * np.abs(ref).max()
    assert np.abs(A.inv_preemphasis(p) - y).max() < 1e-4


def test_griffin_lim_matches_oracle(audio):
    A, hp = audio
    y = _speechlike(30 * 250 + 1000, 6)
    # a spectrogram with real dynamics: under the shipped min_level_db=+100 the
    # features of this signal saturate at 1.0, which Griffin-Lim maps to an all-zero waveform
    spec = AO.spectrogram(y, dict(HP, min_level_db=-100)).T[:31]            # [T, F] in [0, 1]
    assert spec.std() > 0.05
    for iters in (0, 1, 5):
        got = A.griffin_lim_gpu(spec, iters=iters).cpu().numpy()
        ref = AO.inv_spectrogram_tensorflow(spec, HP, iters=iters)
        assert got.shape == ref.shape == ((31 - 1) * 250 + 1000,)
        # relative to the signal's peak; phase normalisation E/|E| is ill-conditioned where |E| ~ 0,
        # so the error grows with the iteration count
        tol = 2e-4 if iters <= 1 else 5e-3
        assert np.abs(got - ref).max() < tol * np.abs(ref).max(), (iters, np.abs(got - ref).max(), np.abs(ref).max())


def test_griffin_lim_batched_and_roundtrip_gain(audio):
    A, hp = audio
    ys = [_speechlike(20 * 250 + 1000, s) for s in (7, 8)]
    specs = np.stack([AO.spectrogram(y, dict(HP, min_level_db=-100)).T[:21] for y in ys])
    out = A.griffin_lim_gpu(specs, iters=3).cpu().numpy()
    for i in range(2):
        single = A.griffin_lim_gpu(specs[i], iters=3).cpu().numpy()
        assert np.array_equal(out[i], single)
    assert np.isfinite(out).all() and np.abs(out).max() > 0
    # saturated input (all ones) -> constant magnitude, zero phase -> impulse at n=0 where Hann is 0
    z = A.griffin_lim_gpu(np.ones((21, 1025), np.float32), iters=2).cpu().numpy()
    assert np.abs(z).max() < 1e-3


def test_griffin_lim_at_the_benchmarked_size(audio):
    """bench.py times 60 iterations on 797 frames (10 s of audio): the same call against the oracle, on the waveform
    after 0, 1 and 60 iterations and on what Griffin-Lim optimises, the magnitude of the TF-convention STFT of the
    result."""
    A, hp = audio
    y = _speechlike(200000, 1234)
    spec = AO.spectrogram(y, dict(HP, min_level_db=-100)).T[:797].copy()
    for iters in (0, 1):
        got = A.griffin_lim_gpu(spec, iters=iters).cpu().numpy()
        ref = AO.inv_spectrogram_tensorflow(spec, HP, iters=iters)
        assert got.shape == ref.shape == (796 * 250 + 1000,)
        assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max(), (iters, np.abs(got - ref).max(), np.abs(ref).max())
    got = A.griffin_lim_gpu(spec).cpu().numpy().astype(np.float64)        # hparams: 60 iterations
    ref = AO.inv_spectrogram_tensorflow(spec, HP)
    # this signal is well conditioned: rounding the oracle's state to float32 at every iteration moves its 60-iteration
    # waveform by 8e-7 of the peak, so the fp32 kernel chain is held to a sample-wise bound as well
    assert np.abs(got - ref).max() < 5e-3 * np.abs(ref).max(), (np.abs(got - ref).max(), np.abs(ref).max())
    Sg, Sr = np.abs(AO.tf_stft(got, 2048, 250, 1000)), np.abs(AO.tf_stft(ref, 2048, 250, 1000))
    rel = np.linalg.norm(Sg - Sr) / np.linalg.norm(Sr)
    assert rel < 1e-3, rel


def test_private_helpers_of_the_reference_module(audio):
    """audio.py:77-171: every `_`-helper the reference's callers could reach, against the float64 oracle."""
    A, hp = audio
    n_fft, hop, win = AO.stft_parameters(HP)
    y = _speechlike(7000, 9)
    # librosa-style pair (features): centred transform and its window-sum normalised inverse
    D = A._stft(y)
    R = AO.librosa_stft(y, n_fft, hop, win)
    assert D.shape == R.shape and D.dtype == np.complex64
    assert np.abs(D - R).max() < 2e-4 * np.abs(R).max()
    back = A._istft(R.astype(np.complex64))
    ref = AO.librosa_istft(R, n_fft, hop, win)
    assert back.shape == ref.shape
    assert np.abs(back - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(back[n_fft:-n_fft] - y[n_fft:len(back) - n_fft]).max() < 1e-4         # perfect reconstruction inside
    # TF-style pair (Griffin-Lim): un-centred, no normalisation; a batch axis like the graph ops
    E = A._stft_tensorflow(y[None, :])
    Rt = AO.tf_stft(y, n_fft, hop, win)
    assert E.shape == (1,) + Rt.shape
    assert np.abs(E[0] - Rt).max() < 2e-4 * np.abs(Rt).max()
    w = A._istft_tensorflow(Rt[None].astype(np.complex64))
    rt = AO.tf_istft(Rt, n_fft, hop, win)
    assert w.shape == (1, len(rt))
    assert np.abs(w[0] - rt).max() < 2e-5 * np.abs(rt).max()
    # element-wise conversions and the mel projection
    x = np.abs(R).astype(np.float32)
    assert np.abs(A._amp_to_db(x) - AO.amp_to_db(x)).max() < 2e-4
    db = np.linspace(-120, 20, 57, dtype=np.float32)
    assert np.abs(A._db_to_amp(db) / np.power(10.0, db * 0.05) - 1).max() < 1e-5
    S = np.linspace(-150, 150, 61, dtype=np.float32)
    assert np.abs(A._normalize(S) - AO.normalize(S, HP)).max() < 1e-6
    u = np.linspace(-0.5, 1.5, 41, dtype=np.float32)
    assert np.abs(A._denormalize(u) - AO.denormalize(u, HP)).max() < 1e-4
    assert np.abs(A._linear_to_mel(x) - AO.mel_basis(HP["sample_rate"], n_fft, HP["num_mels"]) @ x).max() < 1e-4 * x.max()
    # the reference's own composition of the helpers reproduces its public function
    lin = A._normalize(A._amp_to_db(np.abs(A._stft(A.preemphasis(y)))) - hp.ref_level_db)
    assert np.abs(lin - A.spectrogram(y)).max() < 2e-4
    # magnitudes in, waveform out (audio.py:90-103)
    spec = AO.spectrogram(y, dict(HP, min_level_db=-100)).T[:20]
    hp.min_level_db = -100
    try:
        mags = np.power(A._db_to_amp(A._denormalize(spec) + hp.ref_level_db), hp.power).astype(np.float32)
        a = A._griffin_lim_tensorflow(mags)
        b = A.inv_spectrogram_tensorflow(spec)
    finally:
        hp.min_level_db = 100
    assert np.abs(a - b).max() < 2e-3 * np.abs(b).max()


def test_other_transform_sizes_take_the_generic_kernels(audio):
    """num_freq 513 (n_fft 1024), 16 kHz: features, Griffin-Lim (the one-workgroup-per-frame LDS kernel; the
    wave-per-frame kernel is for n_fft 2048 only) and the TF-style transform pair against the oracle."""
    A, hp = audio
    keep = {k: getattr(hp, k) for k in ("num_freq", "sample_rate", "min_level_db")}
    hp.num_freq, hp.sample_rate, hp.min_level_db = 513, 16000, -100
    H2 = dict(HP, num_freq=513, sample_rate=16000, min_level_db=-100)
    try:
        n_fft, hop, win = AO.stft_parameters(H2)
        assert (n_fft, hop, win) == (1024, 200, 800)
        y = _speechlike(9000, 21)
        lin, mel = A.spectrogram_and_mel(y)
        assert np.abs(lin - AO.spectrogram(y, H2)).max() < 2e-4
        assert np.abs(mel - AO.melspectrogram(y, H2)).max() < 2e-4
        spec = AO.spectrogram(y, H2).T[:25].copy()
        for iters in (0, 2):
            got = A.griffin_lim_gpu(spec, iters=iters).cpu().numpy()
            ref = AO.inv_spectrogram_tensorflow(spec, H2, iters=iters)
            assert got.shape == ref.shape == (24 * hop + win,)
            assert np.abs(got - ref).max() < (2e-4 if iters == 0 else 2e-3) * np.abs(ref).max(), iters
        E = A._stft_tensorflow(y)
        Rt = AO.tf_stft(y, n_fft, hop, win)
        assert np.abs(E - Rt).max() < 2e-4 * np.abs(Rt).max()
        assert np.abs(A._istft_tensorflow(Rt.astype(np.complex64)) - AO.tf_istft(Rt, n_fft, hop, win)).max() < 2e-5 * np.abs(y).max() * 10
    finally:
        for k, v in keep.items():
            setattr(hp, k, v)


def test_load_wav_reads_librispeech_flac_like_the_same_pcm_in_a_wav(audio, tmp_path):
    """LibriSpeech: 16 kHz mono .flac, resampled to the model's 20 kHz like any other input (audio.py:13-14)."""
    import os
    import wave
    import flac_writer as FW
    from test_flac_cpu import _lpc
    A, hp = audio
    y = _speechlike(16000, 3)
    pcm = np.round(y / np.abs(y).max() * 0.7 * 32767).astype(np.int64)[:, None]
    frames, pos = [], 0
    while pos < len(pcm):
        size = min(4096, len(pcm) - pos)
        spec = _lpc(pcm[pos:pos + size, 0], 8)
        spec.update(porder=2 if size == 4096 else 0)
        frames.append(dict(size=size, subframes=[spec]))
        pos += size
    pf, pw = os.path.join(str(tmp_path), "a.flac"), os.path.join(str(tmp_path), "a.wav")
    with open(pf, "wb") as f:
        f.write(FW.encode(pcm, 16, 16000, frames))
    with wave.open(pw, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
        f.writeframes(pcm[:, 0].astype("<i2").tobytes())
    a, b = A.load_wav(pf), A.load_wav(pw)
    assert a.shape == b.shape == (int(np.ceil(16000 * hp.sample_rate / 16000)),) and np.array_equal(a, b)


def test_process_utterance_trims_before_the_features(audio, tmp_path):
    """process.py:23-36: features of trim_wav(load_wav(path)) - against the oracle's trim and the oracle's spectrograms;
    the device-cache path of the feeder sees the same frames."""
    from nspeech_amd.datasets import process as P
    A, hp = audio
    rng = np.random.default_rng(3)
    body = _speechlike(30000, 9)
    y = np.concatenate([rng.normal(0, 1e-4, 7000), body / np.abs(body).max() * 0.6, rng.normal(0, 1e-4, 9000)]).astype(np.float32)
    path = str(tmp_path / "utt000.wav")
    A.save_wav(y, path)
    wav = A.load_wav(path)
    ref_wav = AO.trim_wav(wav)
    assert 30000 < len(ref_wav) < len(wav) - 8000
    idx, got_wav, lin, mel, n_frames = P.process_utterance(path)
    assert idx == "utt000" and np.array_equal(got_wav, ref_wav)
    rl, rm = AO.spectrogram(ref_wav, HP), AO.melspectrogram(ref_wav, HP)
    assert lin.shape == rl.T.shape and mel.shape == rm.T.shape and n_frames == rl.shape[1] == 1 + len(ref_wav) // 250
    assert np.abs(lin - rl.T).max() < 2e-4 and np.abs(mel - rm.T).max() < 2e-4
