"""CPU checks of the audio oracle against independent formulations available in this container
(scipy.signal.lfilter / stft, closed-form identities).  No GPU."""
import numpy as np
import scipy.signal

from oracle import audio_oracle as AO

HP = dict(num_mels=80, num_freq=1025, sample_rate=20000, frame_length_ms=50, frame_shift_ms=12.5,
          preemphasis=0.97, min_level_db=100, ref_level_db=20, power=1.5, griffin_lim_iters=60)


def _speechlike(L, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(L) / 20000.0
    f0 = rng.uniform(90, 250)
    y = sum(np.sin(2 * np.pi * f0 * (h + 1) * t) / (h + 1) for h in range(5))
    y *= 0.5 + 0.5 * np.sin(2 * np.pi * 4 * t)
    y += rng.normal(0, 0.01, L)
    return (0.8 * y / np.abs(y).max()).astype(np.float32)


def test_stft_parameters_and_sizes():
    assert AO.stft_parameters(HP) == (2048, 250, 1000)
    y = _speechlike(5000)
    assert AO.spectrogram(y, HP).shape == (1025, 1 + 5000 // 250)
    assert AO.melspectrogram(y, HP).shape == (80, 21)


def test_preemphasis_matches_scipy_lfilter():
    y = _speechlike(3000, 1)
    assert np.allclose(AO.preemphasis(y, 0.97), scipy.signal.lfilter([1, -0.97], [1], y), atol=1e-12)
    assert np.allclose(AO.inv_preemphasis(y, 0.97), scipy.signal.lfilter([1], [1, -0.97], y), atol=1e-9)
    assert np.allclose(AO.inv_preemphasis(AO.preemphasis(y, 0.97), 0.97), y, atol=1e-9)


def test_librosa_stft_matches_scipy_stft():
    # scipy's stft with boundary=None on the reflect-padded signal and a zero-padded Hann is the
    # same framing; scipy scales by 1/sum(window)
    y = _speechlike(4000, 2).astype(np.float64)
    n_fft, hop, win = 2048, 250, 1000
    D = AO.librosa_stft(y, n_fft, hop, win)
    w = np.zeros(n_fft)
    w[(n_fft - win) // 2:(n_fft - win) // 2 + win] = scipy.signal.get_window("hann", win, fftbins=True)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    _, _, Z = scipy.signal.stft(yp, window=w, nperseg=n_fft, noverlap=n_fft - hop, boundary=None, padded=False)
    assert D.shape == Z.shape
    assert np.allclose(D, Z * w.sum(), atol=1e-8)


def test_mel_basis_properties():
    B = AO.mel_basis(20000, 2048, 80)
    assert B.shape == (80, 1025) and (B >= 0).all()
    peaks = B.argmax(axis=1)
    assert (np.diff(peaks) > 0).all()           # monotone centre frequencies
    # Slaney area normalisation: integral of each triangle over Hz is ~1
    df = 10000.0 / 1024
    assert np.allclose(B.sum(axis=1) * df, 1.0, atol=0.05)


def test_tf_istft_gain_and_length():
    n_fft, hop, win = 2048, 250, 1000
    y = _speechlike(10 * 250 + 1000, 3).astype(np.float64)
    S = AO.tf_stft(y, n_fft, hop, win)
    assert S.shape == (11, 1025)
    z = AO.tf_istft(S, n_fft, hop, win)
    assert len(z) == len(y)
    # sum of squared periodic Hann windows at 75 % overlap = 1.5: the un-normalised round-trip gain
    mid = slice(1000, len(y) - 1000)
    assert np.allclose(z[mid], 1.5 * y[mid], atol=1e-9)


def test_normalize_sign_agnostic_q1():
    S = np.array([-120.0, -50.0, 0.0, 30.0])
    hp_pos = dict(HP, min_level_db=100)
    hp_neg = dict(HP, min_level_db=-100)
    assert np.allclose(AO.normalize(S, hp_pos), np.clip(1 - S / 100, 0, 1))
    assert np.allclose(AO.normalize(S, hp_neg), np.clip((S + 100) / 100, 0, 1))
    x = np.array([0.0, 0.25, 1.0])
    assert np.allclose(AO.denormalize(x, hp_pos), 100 - 100 * x)


def test_find_endpoint():
    wav = np.concatenate([0.5 * np.ones(30000), np.zeros(40000)])
    e = AO.find_endpoint(wav, HP)
    assert 30000 <= e <= 30000 + 2 * 4000
    assert AO.find_endpoint(0.5 * np.ones(50000), HP) == 50000


def test_resample_22050_to_20000_against_polyphase_reference(tmp_path):
    """load_wav resamples like librosa.load (resampy 'kaiser_best', restated, parity unpinned): checked against an
    independent band-limited resampler (scipy.signal.resample_poly, 800/882) on in-band material, for length, gain
    and waveform, and through load_wav on a 22 050 Hz PCM16 file."""
    import wave
    from scipy.signal import resample_poly
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.utils import audio
    hp = hparams_mod.load("taco2")
    hparams_mod.set_hparams(hp)
    sr0, sr1 = 22050, hp.sample_rate
    t = np.arange(sr0 // 3) / sr0
    x = (0.4 * np.sin(2 * np.pi * 220 * t) + 0.3 * np.sin(2 * np.pi * 1870 * t + 0.3) + 0.2 * np.sin(2 * np.pi * 6100 * t)).astype(np.float32)
    y = audio.resample(x, sr0, sr1)
    assert len(y) == int(len(x) * sr1 / sr0)
    ref = resample_poly(x.astype(np.float64), 800, 882)[:len(y)]
    mid = slice(1500, len(y) - 1500)                       # away from the edge effects of the two filters
    assert np.abs(y[mid] - ref[mid]).max() < 2e-3
    t1 = np.arange(len(y)) / sr1
    exact = 0.4 * np.sin(2 * np.pi * 220 * t1) + 0.3 * np.sin(2 * np.pi * 1870 * t1 + 0.3) + 0.2 * np.sin(2 * np.pi * 6100 * t1)
    assert np.abs(y[mid] - exact[mid]).max() < 2e-3
    # a tone above the new Nyquist (10 kHz) is removed instead of aliased
    hi = audio.resample(np.sin(2 * np.pi * 10700 * t).astype(np.float32), sr0, sr1)
    assert np.abs(hi[mid]).max() < 2e-2
    path = str(tmp_path / "a.wav")
    with wave.open(path, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr0)
        f.writeframes((x * 32767).astype("<i2").tobytes())
    w = audio.load_wav(path)
    assert len(w) == len(y) and np.abs(w[mid] - y[mid]).max() < 1e-3


def test_librosa_istft_inverts_the_centred_transform():
    """librosa_istft (window-sum normalised overlap-add, centred trim) is the exact inverse of librosa_stft for the
    shipped window / hop (win = 4 hop): the restatement the GPU `_istft` is checked against."""
    n_fft, hop, win = AO.stft_parameters(HP)
    y = _speechlike(6000, 3)
    D = AO.librosa_stft(y, n_fft, hop, win)
    back = AO.librosa_istft(D, n_fft, hop, win)
    assert len(back) == hop * (D.shape[1] - 1)
    assert np.abs(back - y[:len(back)]).max() < 1e-9
