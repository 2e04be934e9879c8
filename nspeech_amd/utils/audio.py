"""Drop-in for neural_speech/utils/audio.py: same functions, NumPy in / NumPy out, hyper-parameters
read from the global get_hparams() at call time - but the DSP runs in hand-written HIP kernels
(csrc/audio.hip) through the C ABI.  Spectrograms are [F, T] exactly like the reference.

Host-side here: wav file I/O, the immutable tables (Hann window, FFT twiddles, Slaney mel basis,
built once per configuration in float64 NumPy and uploaded), and find_endpoint's threshold scan.
"""
import wave

import numpy as np
import torch

from .. import ops
from .. import _lib as L
from ..hparams import get_hparams

_tables = {}


def _dev():
    if not torch.cuda.is_available():
        raise L.NSError("nspeech_amd.utils.audio needs a GPU: the DSP kernels have no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stft_parameters():
    hp = get_hparams()
    n_fft = (hp.num_freq - 1) * 2
    hop_length = int(hp.frame_shift_ms / 1000 * hp.sample_rate)
    win_length = int(hp.frame_length_ms / 1000 * hp.sample_rate)
    return n_fft, hop_length, win_length


def _hz_to_mel(f):
    f = np.atleast_1d(np.asarray(f, np.float64))
    out = f / (200.0 / 3)
    big = f >= 1000.0
    out[big] = 15.0 + np.log(f[big] / 1000.0) * (27.0 / np.log(6.4))
    return out


def _mel_to_hz(m):
    m = np.atleast_1d(np.asarray(m, np.float64))
    out = m * (200.0 / 3)
    big = m >= 15.0
    out[big] = 1000.0 * np.exp((m[big] - 15.0) * (np.log(6.4) / 27.0))
    return out


def _build_mel_basis():
    """librosa.filters.mel(sr, n_fft, n_mels) of librosa 0.6.0: Slaney mel scale, triangular
    filters between n_mels+2 band edges from 0 to sr/2, each scaled by 2/(f[i+2]-f[i])."""
    hp = get_hparams()
    n_fft = (hp.num_freq - 1) * 2
    freqs = np.arange(hp.num_freq, dtype=np.float64) * (hp.sample_rate / float(n_fft))
    edges = _mel_to_hz(np.linspace(_hz_to_mel(0.0)[0], _hz_to_mel(hp.sample_rate / 2.0)[0], hp.num_mels + 2))
    basis = np.zeros((hp.num_mels, hp.num_freq))
    for i in range(hp.num_mels):
        lo, mid, hi = edges[i], edges[i + 1], edges[i + 2]
        up = (freqs - lo) / (mid - lo)
        down = (hi - freqs) / (hi - mid)
        basis[i] = np.maximum(0.0, np.minimum(up, down)) * (2.0 / (hi - lo))
    return basis


def _get_tables():
    _dev()
    hp = get_hparams()
    n_fft, hop, win = _stft_parameters()
    key = (n_fft, win, hp.num_mels, hp.sample_rate, torch.cuda.current_device())
    if key not in _tables:
        dev = _dev()
        w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win) / win)
        m = np.arange(n_fft // 2)
        tw = np.stack([np.cos(2 * np.pi * m / n_fft), -np.sin(2 * np.pi * m / n_fft)], axis=1)
        _tables[key] = dict(
            window=torch.tensor(w, dtype=torch.float32, device=dev),
            twiddle=torch.tensor(tw, dtype=torch.float32, device=dev).contiguous(),
            mel_basis=torch.tensor(_build_mel_basis(), dtype=torch.float32, device=dev).contiguous())
    return _tables[key]


# ---------------------------------------------------------------- wav I/O (host)
_RESAMPLE_FILTER = None


def _kaiser_best():
    """resampy's 'kaiser_best' interpolation table [3P, resampy filters.py]: rolloff * sinc(rolloff * t) for t in
    [0, num_zeros] at 2^precision points per zero crossing, tapered by the right half of a Kaiser window."""
    global _RESAMPLE_FILTER
    if _RESAMPLE_FILTER is None:
        num_zeros, precision, rolloff, beta = 64, 9, 0.9475937167399596, 14.769656459379492
        num_bits = 2 ** precision
        n = num_bits * num_zeros
        sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
        taper = np.kaiser(2 * n + 1, beta)[n:]
        _RESAMPLE_FILTER = (taper * sinc_win, num_bits)
    return _RESAMPLE_FILTER


def resample(x, sr_orig, sr_new):
    """librosa.core.load's default resampler (librosa 0.6.0 -> resampy.resample(..., filter='kaiser_best'), audio.py:14)
    restated [3P, parity unpinned]: band-limited sinc interpolation, the table read with linear interpolation between
    entries, gain and cut-off scaled by the ratio when down-sampling.  Vectorised over the output samples."""
    # float64 tensor arithmetic: on the GPU when there is one (a 10 s clip takes milliseconds), else on the host
    dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    x = torch.as_tensor(np.asarray(x, np.float64), device=dev)
    ratio = float(sr_new) / float(sr_orig)
    n_out = int(x.shape[0] * ratio)
    win_np, num_table = _kaiser_best()
    win = torch.as_tensor(win_np * (ratio if ratio < 1 else 1.0), device=dev)
    delta = torch.zeros_like(win)
    delta[:-1] = win[1:] - win[:-1]
    scale = min(1.0, ratio)
    index_step = int(scale * num_table)
    nwin, n_orig = win.shape[0], x.shape[0]
    tr = torch.arange(n_out, device=dev, dtype=torch.float64) * (1.0 / ratio)     # time register
    n = tr.long()
    y = torch.zeros(n_out, dtype=torch.float64, device=dev)
    zero = torch.zeros((), dtype=torch.long, device=dev)

    def wing(frac, count, sign, base):
        nonlocal y
        idx = frac * num_table
        off = idx.long()
        eta = idx - off
        cmax = torch.minimum(count, (nwin - off) // index_step)
        for i in range(int(cmax.max().item())):
            ok = i < cmax
            k = torch.where(ok, off + i * index_step, zero)
            w = win[k] + eta * delta[k]
            y = y + torch.where(ok, w * x[torch.where(ok, base + sign * i, zero)], torch.zeros_like(y))
    frac = scale * (tr - n)
    wing(frac, n + 1, -1, n)                          # left wing: x[n], x[n-1], ...
    wing(scale - frac, n_orig - n - 1, +1, n + 1)     # right wing: x[n+1], x[n+2], ...
    return y.float().cpu().numpy()


def _strip_id3(data):
    """Tags that taggers wrap around a FLAC stream: a leading ID3v2 block (10-byte header, sync-safe size, optional
    footer) and a trailing 128-byte ID3v1 block."""
    if data[:3] == b"ID3" and len(data) > 10:
        size = ((data[6] & 0x7f) << 21) | ((data[7] & 0x7f) << 14) | ((data[8] & 0x7f) << 7) | (data[9] & 0x7f)
        data = data[10 + size + (10 if data[5] & 0x10 else 0):]
    if len(data) > 128 and data[-128:-125] == b"TAG":
        data = data[:-128]
    return data


MAX_FLAC_SECONDS = 3600      # longest stream _load_flac decodes (an unbounded retry loop could be OOM-killed)


def _load_flac(path):
    """FLAC file -> (float32 [samples, channels] in [-1, 1), sample rate): ns_flac_decode (csrc/flac.hip, host code),
    checked against the MD5 of the decoded PCM that the encoder stored in STREAMINFO (all-zero = not stored)."""
    import ctypes as C
    import hashlib
    data = _strip_id3(open(path, "rb").read())
    lib = L.lib()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    sr, ch, bps, total = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
    md5 = (C.c_uint8 * 16)()
    L.check(lib.ns_flac_info(buf, C.c_size_t(len(data)), C.byref(sr), C.byref(ch), C.byref(bps), C.byref(total), md5), "ns_flac_info")
    # no total in STREAMINFO: start from a modest guess and double on NS_ERR_SHORT_BUFFER.  No bound follows from the
    # file size (a CONSTANT subframe codes a whole block in a few bytes), so a damaged or crafted stream could ask for
    # gigabytes: the decoded length is capped at MAX_FLAC_SECONDS of audio (an utterance corpus has nothing near it).
    limit = int(MAX_FLAC_SECONDS * max(sr.value, 1))
    if total.value > limit:
        raise ValueError("%s: STREAMINFO announces %d samples, more than the %d s this loader accepts" % (path, total.value, MAX_FLAC_SECONDS))
    cap = total.value if total.value > 0 else min(max(1 << 16, len(data)), limit)
    got = C.c_int64()
    while True:
        out = np.empty((cap, ch.value), dtype=np.int32)
        rc = lib.ns_flac_decode(buf, C.c_size_t(len(data)), out.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(cap), C.byref(got))
        if rc == L.NS_ERR_SHORT_BUFFER and total.value == 0:
            if cap >= limit:
                raise ValueError("%s: more than %d s of audio in a stream without a sample count" % (path, MAX_FLAC_SECONDS))
            cap = min(2 * cap, limit)
            continue
        L.check(rc, "ns_flac_decode")
        break
    out = out[:got.value]
    if total.value > 0 and got.value != total.value:
        raise ValueError("%s: %d samples decoded, STREAMINFO announces %d" % (path, got.value, total.value))
    if any(md5):
        nb = (bps.value + 7) // 8
        pcm = out.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :nb].tobytes()
        if hashlib.md5(pcm).digest() != bytes(md5):
            raise ValueError("%s: decoded audio does not match the MD5 in STREAMINFO" % path)
    return (out.astype(np.float32) / float(1 << (bps.value - 1))), sr.value


def load_wav(path, offset=0.0, duration=None):
    """PCM16 / float32 RIFF or FLAC reader + mono mix-down + resampling to hparams.sample_rate, as librosa.core.load
    does for the reference (audio.py:13-14; LJSpeech is 22 050 Hz WAV, LibriSpeech 16 000 Hz FLAC, audio.yaml asks for
    20 000 Hz)."""
    with open(path, "rb") as f:
        magic = f.read(4)
        if magic[:3] == b"ID3":              # an ID3v2 tag in front of the stream: look behind it
            f.seek(0)
            magic = _strip_id3(f.read())[:4]
    hp = get_hparams()
    if magic == b"fLaC":                     # LibriSpeech (datasets/corpus/ljspeech.py:17)
        x, sr = _load_flac(path)
        ch = x.shape[1]
        x = x.reshape(-1) if ch == 1 else x
    else:
        try:
            with wave.open(path, "rb") as f:
                sr, n, width, ch = f.getframerate(), f.getnframes(), f.getsampwidth(), f.getnchannels()
                raw = f.readframes(n)
        except wave.Error as e:
            raise ValueError("%s: only RIFF/WAV and FLAC input is supported (%s)" % (path, e))
        if width == 2:
            x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        elif width == 4:
            x = np.frombuffer(raw, dtype="<f4").astype(np.float32)
        else:
            raise ValueError("unsupported sample width %d" % width)
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1)
    s = int(offset * sr)                     # librosa seeks / truncates at the native rate, then resamples
    e = None if duration is None else s + int(duration * sr)
    x = x[s:e]
    if sr != hp.sample_rate:
        x = resample(x, sr, hp.sample_rate)
    return x


def save_wav(wav, path):
    """audio.py:17-19 scales to +-32767 (in place in the reference); written as int16 PCM."""
    wav = np.asarray(wav, np.float64) * (32767 / max(0.01, np.max(np.abs(wav))))
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(get_hparams().sample_rate)
        f.writeframes(wav.astype("<i2").tobytes())


def load_spectrogram(path):
    spec = np.load(path)
    return spec, spec.shape[1]


def save_spectrogram(spec, path):
    np.save(path, spec, allow_pickle=False)


# ---------------------------------------------------------------- filters
def _preemph(x, inverse):
    xt = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(_dev())
    y = torch.empty_like(xt)
    p = L.struct("ns_preemphasis_params")
    p.x, p.y, p.n, p.coef, p.inverse = ops.ptr(xt), ops.ptr(y), xt.numel(), float(get_hparams().preemphasis), inverse
    L.call("ns_preemphasis", p, ops.stream())
    return y.cpu().numpy()


def preemphasis(x):
    return _preemph(x, 0)


def inv_preemphasis(x):
    return _preemph(x, 1)


# ---------------------------------------------------------------- spectrograms
def _spectrograms(y, want_lin, want_mel, want_stft=False, on_device=False):
    hp = get_hparams()
    n_fft, hop, win = _stft_parameters()
    tb = _get_tables()
    dev = _dev()
    wav = torch.as_tensor(np.ascontiguousarray(y, dtype=np.float32)).to(dev)
    Lw = wav.numel()
    T = 1 + Lw // hop
    lin = torch.empty(T * hp.num_freq, dtype=torch.float32, device=dev) if want_lin else None
    mel = torch.empty(T * hp.num_mels, dtype=torch.float32, device=dev) if want_mel else None
    p = L.struct("ns_spectrogram_params")
    p.wav, p.L, p.preemph = ops.ptr(wav), Lw, 0.0 if want_stft else float(hp.preemphasis)
    if want_stft:           # the transform alone (_stft): no pre-emphasis, complex bins out
        cx = torch.empty(T * hp.num_freq * 2, dtype=torch.float32, device=dev)
        p.stft_out = ops.ptr(cx)
    p.n_fft, p.hop, p.win, p.T = n_fft, hop, win, T
    p.window, p.twiddle = ops.ptr(tb["window"]), ops.ptr(tb["twiddle"])
    p.mel_basis, p.n_mels = ops.ptr(tb["mel_basis"]), hp.num_mels
    p.ref_level_db, p.min_level_db = float(hp.ref_level_db), float(hp.min_level_db)
    p.lin_out, p.mel_out = ops.ptr(lin), ops.ptr(mel)
    L.call("ns_spectrogram", p, ops.stream())
    if want_stft:
        return torch.view_as_complex(cx.view(T, hp.num_freq, 2)).t().cpu().numpy()
    if on_device:
        return lin, mel
    out_lin = lin.view(T, hp.num_freq).t().cpu().numpy() if want_lin else None
    out_mel = mel.view(T, hp.num_mels).t().cpu().numpy() if want_mel else None
    return out_lin, out_mel


def spectrogram(y):
    return np.ascontiguousarray(_spectrograms(y, True, False)[0])


def melspectrogram(y):
    return np.ascontiguousarray(_spectrograms(y, False, True)[1])


def spectrogram_and_mel(y):
    """Both features from ONE STFT (the reference recomputes the STFT for each, process.py:28-33)."""
    a, b = _spectrograms(y, True, True)
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def spectrogram_and_mel_device(y):
    """The same two features left where the kernel wrote them: (linear [T, F], mel [T, M]) float32 CUDA tensors in the
    time-major layout a training batch uses - for the feeder's HBM-resident feature cache (datasets/datafeeder.py)."""
    hp = get_hparams()
    lin, mel = _spectrograms(y, True, True, on_device=True)
    return lin.view(-1, hp.num_freq), mel.view(-1, hp.num_mels)


def griffin_lim_gpu(spec, iters=None, raw_magnitude=False):
    """spec: torch CUDA tensor or array [T, F] / [N, T, F] (normalised; with raw_magnitude the magnitudes S^power
    themselves).  Returns a CUDA tensor [L] / [N, L], L = (T-1)*hop + win, before inv_preemphasis (audio.py:51-58)."""
    hp = get_hparams()
    n_fft, hop, win = _stft_parameters()
    tb = _get_tables()
    dev = _dev()
    st = spec if torch.is_tensor(spec) else torch.as_tensor(np.asarray(spec, dtype=np.float32))
    st = st.to(dev, torch.float32).contiguous()
    single = st.dim() == 2
    if single:
        st = st.unsqueeze(0)
    N, T, F = st.shape
    assert F == hp.num_freq
    Lout = (T - 1) * hop + win
    wav = torch.empty(N * Lout, dtype=torch.float32, device=dev)
    p = L.struct("ns_griffin_lim_params")
    p.spec, p.N, p.T = ops.ptr(st), N, T
    p.n_fft, p.hop, p.win = n_fft, hop, win
    p.iters = int(hp.griffin_lim_iters if iters is None else iters)
    p.power, p.ref_level_db, p.min_level_db = float(hp.power), float(hp.ref_level_db), float(hp.min_level_db)
    p.window, p.twiddle, p.wav = ops.ptr(tb["window"]), ops.ptr(tb["twiddle"]), ops.ptr(wav)
    p.raw_magnitude = int(bool(raw_magnitude))
    nbytes = L.lib().ns_griffin_lim_work_bytes
    nbytes.restype = __import__("ctypes").c_size_t
    work = torch.empty((nbytes(__import__("ctypes").byref(p)) + 3) // 4, dtype=torch.float32, device=dev)
    p.work = ops.ptr(work)
    L.call("ns_griffin_lim", p, ops.stream())
    out = wav.view(N, Lout)
    return out[0] if single else out


def inv_spectrogram_tensorflow(spectrogram):
    """Name kept from the reference (audio.py:51): Griffin-Lim on the GPU; [T,F] in, waveform out,
    pre-emphasis NOT inverted (the caller applies inv_preemphasis, as in the reference)."""
    out = griffin_lim_gpu(spectrogram)
    return out if torch.is_tensor(spectrogram) else out.cpu().numpy()


def inv_spectrogram(spectrogram):
    """audio.py:45-48 takes [F, T]; the live path of the reference is the TF variant,
    so this uses the same deterministic zero-phase Griffin-Lim, then inverts the pre-emphasis."""
    wav = griffin_lim_gpu(np.asarray(spectrogram, dtype=np.float32).T).cpu().numpy()
    return inv_preemphasis(wav)


# ---------------------------------------------------------------- the reference's private helpers (audio.py:77-171)
# Same names, NumPy in / NumPy out (the *_tensorflow forms also take and then return CUDA tensors), each one a HIP call.
def _pointwise(x, mode):
    is_t = torch.is_tensor(x)
    xt = (x if is_t else torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32))).to(_dev(), torch.float32).contiguous()
    y = torch.empty_like(xt)
    p = L.struct("ns_audio_pointwise_params")
    p.x, p.y, p.n, p.mode, p.min_level_db = ops.ptr(xt), ops.ptr(y), xt.numel(), mode, float(get_hparams().min_level_db)
    L.call("ns_audio_pointwise", p, ops.stream())
    return y if is_t else y.cpu().numpy()


def _amp_to_db(x):
    return _pointwise(x, 0)


def _db_to_amp(x):
    return _pointwise(x, 1)


def _normalize(S):
    return _pointwise(S, 2)


def _denormalize(S):
    return _pointwise(S, 3)


_db_to_amp_tensorflow = _db_to_amp
_denormalize_tensorflow = _denormalize


def _linear_to_mel(spectrogram):
    """audio.py:138-142: mel_basis [n_mels, F] . spectrogram [F, T] (one fp32 product)."""
    tb = _get_tables()
    S = torch.as_tensor(np.ascontiguousarray(spectrogram, dtype=np.float32)).to(_dev())
    F, T = S.shape
    n_mels = tb["mel_basis"].numel() // F
    out = torch.empty(n_mels * T, dtype=torch.float32, device=S.device)
    keep, ops.F32_PASSES = ops.F32_PASSES, 0
    try:
        ops.gemm(tb["mel_basis"], S.reshape(-1), out, n_mels, T, F, F, T, T, b_mode=1)
    finally:
        ops.F32_PASSES = keep
    return out.view(n_mels, T).cpu().numpy()


def _stft(y):
    """audio.py:106-108: librosa.stft(y, n_fft, hop, win) - centred, reflect-padded; complex64 [F, T]."""
    return _spectrograms(y, False, False, want_stft=True)


def _complex_rows(x):
    """complex array [..., F] -> float32 CUDA tensor [..., F, 2]"""
    a = np.ascontiguousarray(np.asarray(x).astype(np.complex64))
    return torch.view_as_real(torch.from_numpy(a)).to(_dev()).contiguous()


def _istft_call(spec_tf, center):
    """spec_tf: CUDA float32 [T, F, 2]."""
    n_fft, hop, win = _stft_parameters()
    tb = _get_tables()
    T = spec_tf.shape[0]
    Lout = (T - 1) * hop + (0 if center else win)
    wav = torch.zeros(max(Lout, 1), dtype=torch.float32, device=spec_tf.device)
    work = torch.empty(max(T * win, 1), dtype=torch.float32, device=spec_tf.device)
    p = L.struct("ns_istft_params")
    p.spec, p.T, p.n_fft, p.hop, p.win, p.center = ops.ptr(spec_tf), T, n_fft, hop, win, int(center)
    p.window, p.twiddle, p.wav, p.work = ops.ptr(tb["window"]), ops.ptr(tb["twiddle"]), ops.ptr(wav), ops.ptr(work)
    L.call("ns_istft", p, ops.stream())
    return wav[:Lout]


def _istft(y):
    """audio.py:111-113: librosa.istft(D [F, T], hop, win): window-sum normalised overlap-add, centred trim."""
    return _istft_call(_complex_rows(np.asarray(y).T), True).cpu().numpy()


def _stft_tensorflow(signals):
    """audio.py:116-118: tf.contrib.signal.stft(signals [B, L], win, hop, n_fft, pad_end=False) -> complex64 [B, T, F]."""
    n_fft, hop, win = _stft_parameters()
    tb = _get_tables()
    is_t = torch.is_tensor(signals)
    x = (signals if is_t else torch.as_tensor(np.ascontiguousarray(signals, dtype=np.float32))).to(_dev(), torch.float32)
    single = x.dim() == 1
    x = x.reshape(-1, x.shape[-1]).contiguous()
    B, Lw = x.shape
    T = max(0, 1 + (Lw - win) // hop)
    F = n_fft // 2 + 1
    out = torch.zeros(B, T, F, 2, dtype=torch.float32, device=x.device)
    for b in range(B):
        p = L.struct("ns_stft_tf_params")
        p.wav, p.L, p.n_fft, p.hop, p.win, p.T = ops.ptr(x, b * Lw), Lw, n_fft, hop, win, T
        p.window, p.twiddle, p.out = ops.ptr(tb["window"]), ops.ptr(tb["twiddle"]), ops.ptr(out, b * T * F * 2)
        L.call("ns_stft_tf", p, ops.stream())
    c = torch.view_as_complex(out)
    c = c[0] if single else c
    return c if is_t else c.cpu().numpy()


def _istft_tensorflow(stfts):
    """audio.py:121-123: tf.contrib.signal.inverse_stft(stfts [B, T, F], win, hop, n_fft) -> [B, (T-1) hop + win]."""
    is_t = torch.is_tensor(stfts)
    x = torch.view_as_real(stfts.to(torch.complex64)).to(_dev()).contiguous() if is_t else _complex_rows(stfts)
    single = x.dim() == 3
    x = x.reshape(-1, x.shape[-3], x.shape[-2], 2)
    out = torch.stack([_istft_call(x[b].contiguous(), False) for b in range(x.shape[0])])
    out = out[0] if single else out
    return out if is_t else out.cpu().numpy()


def _griffin_lim_tensorflow(S):
    """audio.py:90-103: S [T, F] = the magnitudes (already denormalised, dB -> amplitude, ^power)."""
    out = griffin_lim_gpu(S, raw_magnitude=True)
    return out if torch.is_tensor(S) else out.cpu().numpy()


def _griffin_lim(S):
    """audio.py:77-87 takes [F, T].  The reference draws a random initial phase (and uses np.complex, gone from NumPy
    1.24: dead code there); this is the deterministic zero-phase TF variant on the same magnitudes."""
    return griffin_lim_gpu(np.asarray(S, dtype=np.float32).T, raw_magnitude=True).cpu().numpy()


def find_endpoint(wav, threshold_db=-40, min_silence_sec=0.8):
    hp = get_hparams()
    window_length = int(hp.sample_rate * min_silence_sec)
    hop_length = int(window_length / 4)
    threshold = np.power(10.0, threshold_db * 0.05)
    for x in range(hop_length, len(wav) - window_length, hop_length):
        if np.max(wav[x:x + window_length]) < threshold:
            return x + hop_length
    return len(wav)
