"""CPU oracle for the simple WaveNet (neural_speech/models/wavenet_simple.py).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic lives in tensorflow-gpu==1.7.0 (tf.nn.conv1d, softmax_cross_entropy_with_logits),
which cannot be imported here, and the reference holds no fixtures for this path.  Restated in PyTorch-CPU float64
with autograd, each block citing the reference lines; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import it.

Shapes follow the reference: audio [N, T] -> mu-law ids [N, T]; the network sees ids[:, :T-1] one-hot; every VALID
causal convolution of dilation d shortens the series by d; the loss compares the last T - receptive_field logits
with ids[:, receptive_field:].  Global / local conditioning and biases are off in the shipped wavenet.yaml and are
not restated."""
import numpy as np
import torch


def dilations(hp):
    """wavenet_simple.py:107."""
    return [2 ** i for _ in range(hp["dilations_depth"]) for i in range(hp["dilations_length"])]


def receptive_field(hp):
    """wavenet_simple.py:124-128."""
    fw = hp["filter_width"]
    return (fw - 1) * sum(dilations(hp)) + 1 + (fw - 1)


def mu_law_encode(audio, q):
    """wavenet_simple.py:586-597 (float32 arithmetic as in the TF graph)."""
    a = np.asarray(audio, np.float32)
    mu = np.float32(q - 1)
    mag = np.log1p(mu * np.minimum(np.abs(a), np.float32(1.0))) / np.log1p(mu)
    sig = np.sign(a) * mag
    return ((sig + 1) / 2 * mu + np.float32(0.5)).astype(np.int32)


def mu_law_decode(ids, q):
    """wavenet_simple.py:600-608."""
    mu = q - 1
    sig = 2 * (np.asarray(ids, np.float32) / mu) - 1
    mag = (1 / mu) * ((1 + mu) ** np.abs(sig) - 1)
    return (np.sign(sig) * mag).astype(np.float32)


def causal_conv(x, w, d):
    """wavenet_simple.py:551-583: VALID conv1d with dilation d; x [N,T,Cin], w [k,Cin,Cout] -> [N, T-(k-1)d, Cout]."""
    k = w.shape[0]
    T = x.shape[1]
    out_w = T - (k - 1) * d
    y = 0
    for j in range(k):
        y = y + x[:, j * d:j * d + out_w, :] @ w[j]
    return y


def network(p, hp, ids):
    """wavenet_simple.py:344-383 on integer inputs ids [N, T0]; returns logits [N, T0 - rf + 1, Q]."""
    q = hp["quantization_channels"]
    x = torch.nn.functional.one_hot(ids.long(), q).to(p["wavenet/causal_layer/filter"].dtype)      # _one_hot :385-397
    cur = causal_conv(x, p["wavenet/causal_layer/filter"], 1)                                          # :246-252
    out_w = ids.shape[1] - receptive_field(hp) + 1
    skips = 0
    for i, d in enumerate(dilations(hp)):
        pre = "wavenet/dilated_stack/layer%d/" % i
        f = causal_conv(cur, p[pre + "filter"], d)                                                     # :288-292
        g = causal_conv(cur, p[pre + "gate"], d)
        out = torch.tanh(f) * torch.sigmoid(g)                                                         # :325
        transformed = out @ p[pre + "dense"][0]                                                        # :328-330
        skips = skips + out[:, out.shape[1] - out_w:, :] @ p[pre + "skip"][0]                          # :332-336
        cur = cur[:, cur.shape[1] - transformed.shape[1]:, :] + transformed                            # :342-344
    t1 = torch.relu(skips)                                                                             # :369-380
    c1 = t1 @ p["wavenet/postprocessing/postprocess1"][0]
    return torch.relu(c1) @ p["wavenet/postprocessing/postprocess2"][0]


def loss(p, hp, ids):
    """initialize + add_loss (wavenet_simple.py:455-502): ids [N, T] mu-law codes of the training clip."""
    logits = network(p, hp, ids[:, :-1])
    target = ids[:, receptive_field(hp):].long()
    return torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), target.reshape(-1)), logits


def predict_proba(p, hp, ids):
    """wavenet_simple.py:436-453: distribution of the next sample after the waveform ids [T] (batch 1)."""
    logits = network(p, hp, ids[None, :])
    return torch.softmax(logits[0, -1].double(), dim=0)


def generate(p, hp, seed_ids, uniforms):
    """Sample-by-sample generation with the full network on a sliding window of receptive_field samples
    (generate_wavenet.py:100-140 without fast generation).  The categorical draw is the inverse CDF of predict_proba
    at the given uniform numbers (np.random.choice does the same with its own stream)."""
    rf = receptive_field(hp)
    wave = list(int(v) for v in seed_ids)
    assert len(wave) >= rf, "seed shorter than the receptive field"
    for u in uniforms:
        pr = predict_proba(p, hp, torch.tensor(wave[-rf:])).numpy()
        c = np.cumsum(pr)
        wave.append(int(min(np.searchsorted(c, u * c[-1], side="right"), len(pr) - 1)))
    return np.asarray(wave, np.int32)
