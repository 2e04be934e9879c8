"""CPU oracle for the audio path of MLCogUP/nspeech (neural_speech/utils/audio.py).  TEST
INFRASTRUCTURE ONLY - never imported by the product path.

PARITY UNPINNED: the reference delegates to librosa==0.6.0, scipy==1.0.0 and
tf.contrib.signal (TF 1.7), none of which is under /root/reference or installable here, and it
ships no audio fixtures.  The functions below restate those libraries' published behaviour in
float64 NumPy (numpy.fft), one per reference function, citing the line they follow.  Where an
independent formulation exists in this container (scipy.signal.lfilter / stft) the tests
cross-check against it.
"""
import numpy as np


def stft_parameters(hp):
    """audio.py:126-130."""
    n_fft = (hp["num_freq"] - 1) * 2
    hop = int(hp["frame_shift_ms"] / 1000 * hp["sample_rate"])
    win = int(hp["frame_length_ms"] / 1000 * hp["sample_rate"])
    return n_fft, hop, win


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True) == tf.contrib.signal.hann_window(n, periodic=True)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def preemphasis(x, coef):
    """audio.py:31-32: lfilter([1,-c],[1],x), zero initial state."""
    x = np.asarray(x, np.float64)
    y = x.copy()
    y[1:] -= coef * x[:-1]
    return y


def inv_preemphasis(x, coef):
    """audio.py:35-36: lfilter([1],[1,-c],x): y[n] = x[n] + c*y[n-1]."""
    x = np.asarray(x, np.float64)
    y = np.empty_like(x)
    acc = 0.0
    for i in range(len(x)):
        acc = x[i] + coef * acc
        y[i] = acc
    return y


def librosa_stft(y, n_fft, hop, win):
    """librosa 0.6.0 stft(center=True, window='hann', pad_mode='reflect'): periodic Hann(win)
    zero-padded centred to n_fft, signal reflect-padded by n_fft//2, frames every hop,
    T = 1 + len(y)//hop.  Returns complex [1 + n_fft//2, T]."""
    y = np.asarray(y, np.float64)
    w = np.zeros(n_fft)
    lpad = (n_fft - win) // 2
    w[lpad:lpad + win] = hann_periodic(win)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    T = 1 + (len(yp) - n_fft) // hop
    frames = np.stack([yp[t * hop:t * hop + n_fft] * w for t in range(T)], axis=1)
    return np.fft.rfft(frames, axis=0)


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr, n_fft, n_mels):
    """librosa 0.6.0 filters.mel(sr, n_fft, n_mels): fmin 0, fmax sr/2, Slaney scale, norm=1."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return weights * enorm[:, None]


def amp_to_db(x):
    return 20 * np.log10(np.maximum(1e-5, x))


def normalize(S, hp):
    """audio.py:162-163 (sign-agnostic in min_level_db; the shipped YAML has +100)."""
    return np.clip((S - hp["min_level_db"]) / -hp["min_level_db"], 0, 1)


def denormalize(S, hp):
    return np.clip(S, 0, 1) * -hp["min_level_db"] + hp["min_level_db"]


def spectrogram(y, hp):
    """audio.py:39-42 -> [num_freq, T] float32."""
    n_fft, hop, win = stft_parameters(hp)
    D = librosa_stft(preemphasis(y, hp["preemphasis"]), n_fft, hop, win)
    S = amp_to_db(np.abs(D)) - hp["ref_level_db"]
    return normalize(S, hp).astype(np.float32)


def melspectrogram(y, hp):
    """audio.py:61-64 -> [num_mels, T] float32 (no ref-level subtraction)."""
    n_fft, hop, win = stft_parameters(hp)
    D = librosa_stft(preemphasis(y, hp["preemphasis"]), n_fft, hop, win)
    B = mel_basis(hp["sample_rate"], n_fft, hp["num_mels"])
    S = amp_to_db(B @ np.abs(D))
    return normalize(S, hp).astype(np.float32)


def librosa_istft(D, n_fft, hop, win):
    """librosa 0.6.0 istft(stft_matrix [F, T], hop_length, win_length) [3P, SURVEY appendix C]: the periodic Hann window
    centred in the n_fft frame times irfft(frame), overlap-added every hop, divided by the summed squared window where
    that exceeds the smallest normal float32, n_fft // 2 trimmed from both ends."""
    D = np.asarray(D)
    T = D.shape[1]
    lpad = (n_fft - win) // 2
    w = np.zeros(n_fft)
    w[lpad:lpad + win] = hann_periodic(win)
    y = np.zeros(n_fft + hop * (T - 1))
    ws = np.zeros_like(y)
    for t in range(T):
        y[t * hop:t * hop + n_fft] += w * np.fft.irfft(D[:, t], n=n_fft)
        ws[t * hop:t * hop + n_fft] += w * w
    nz = ws > np.finfo(np.float32).tiny
    y[nz] /= ws[nz]
    return y[n_fft // 2:len(y) - n_fft // 2]


def tf_stft(y, n_fft, hop, win):
    """tf.contrib.signal.stft(y, win, hop, n_fft, pad_end=False): frames of `win` every `hop`,
    periodic Hann, rfft zero-padded AT THE END to n_fft.  Returns [T, 1+n_fft//2]."""
    y = np.asarray(y, np.float64)
    T = 1 + (len(y) - win) // hop
    w = hann_periodic(win)
    frames = np.stack([y[t * hop:t * hop + win] * w for t in range(T)], axis=0)
    return np.fft.rfft(frames, n=n_fft, axis=1)


def tf_istft(S, n_fft, hop, win):
    """tf.contrib.signal.inverse_stft(S, win, hop, n_fft) with the default window_fn:
    irfft(n_fft)[:win] * Hann, overlap-add, NO window-sum normalisation."""
    T = S.shape[0]
    w = hann_periodic(win)
    frames = np.fft.irfft(S, n=n_fft, axis=1)[:, :win] * w
    y = np.zeros((T - 1) * hop + win)
    for t in range(T):
        y[t * hop:t * hop + win] += frames[t]
    return y


def griffin_lim_tf(S, n_fft, hop, win, iters):
    """audio.py:90-103: zero initial phase, `iters` x { E = STFT(y); y = ISTFT(S * E/max(1e-8,|E|)) }."""
    Sc = S.astype(np.complex128)
    y = tf_istft(Sc, n_fft, hop, win)
    for _ in range(iters):
        est = tf_stft(y, n_fft, hop, win)
        angles = est / np.maximum(1e-8, np.abs(est))
        y = tf_istft(Sc * angles, n_fft, hop, win)
    return y


def inv_spectrogram_tensorflow(spec, hp, iters=None):
    """audio.py:51-58.  spec [T, num_freq] normalised.  Returns the waveform BEFORE inv_preemphasis."""
    n_fft, hop, win = stft_parameters(hp)
    S = np.power(10.0, (denormalize(np.asarray(spec, np.float64), hp) + hp["ref_level_db"]) * 0.05)
    S = np.power(S, hp["power"])
    return griffin_lim_tf(S, n_fft, hop, win, hp["griffin_lim_iters"] if iters is None else iters)


def find_endpoint(wav, hp, threshold_db=-40, min_silence_sec=0.8):
    """audio.py:67-74 (thresholds np.max, not abs: SURVEY Q13)."""
    window_length = int(hp["sample_rate"] * min_silence_sec)
    hop_length = int(window_length / 4)
    threshold = np.power(10.0, threshold_db * 0.05)
    for x in range(hop_length, len(wav) - window_length, hop_length):
        if np.max(wav[x:x + window_length]) < threshold:
            return x + hop_length
    return len(wav)


# ---------------------------------------------------------------------------------------------------------------------
# utterance processing: silence trimming in front of the feature extraction (datasets/process.py:27,39-42,56-68)

def effects_split(y, top_db, frame_length, hop_length):
    """[3P] librosa 0.6.0 `effects.split(y, top_db, frame_length=, hop_length=)` as process.py:41 calls it (ref = np.max):
    feature.rmse on centred, reflect-padded frames -> mean square per frame -> power_to_db relative to the loudest frame
    (amin 1e-10 on both sides, no top_db clamp) -> frames louder than -top_db -> runs of such frames as sample intervals
    [first frame * hop, (last frame + 1) * hop) clipped to the signal.  Plain per-frame loops: this is the checker."""
    y = np.asarray(y, np.float64)
    n = len(y)
    yp = np.pad(y, frame_length // 2, mode="reflect")
    n_frames = 1 + (len(yp) - frame_length) // hop_length
    mse = np.empty(n_frames)
    for i in range(n_frames):
        fr = yp[i * hop_length:i * hop_length + frame_length]
        mse[i] = np.mean(fr * fr)
    amin = 1e-10
    db = 10.0 * np.log10(np.maximum(amin, mse)) - 10.0 * np.log10(max(amin, float(mse.max())))
    loud = db > -top_db
    out, start = [], None
    for i in range(n_frames):
        if loud[i] and start is None:
            start = i
        if not loud[i] and start is not None:
            out.append((start, i))
            start = None
    if start is not None:
        out.append((start, n_frames))
    return [(min(a * hop_length, n), min(b * hop_length, n)) for a, b in out]


def find_start(splits, min_samples=2000):
    """process.py:56-60: the first interval longer than min_samples, moved min_samples to the left."""
    for a, b in splits:
        if b - a > min_samples:
            return max(0, a - min_samples)
    return 0


def find_end(splits, num_samples, min_samples=2000):
    """process.py:63-67: the last interval longer than min_samples, moved min_samples to the right."""
    for a, b in reversed(splits):
        if b - a > min_samples:
            return min(num_samples, b + min_samples)
    return num_samples


def trim_wav(wav, threshold_db=25):
    """process.py:39-42."""
    splits = effects_split(wav, threshold_db, 1024, 512)
    return wav[find_start(splits):find_end(splits, len(wav))]
