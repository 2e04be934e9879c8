"""CPU oracle for the simple WaveNet (neural_speech/models/wavenet_simple.py) and for the options of the full
WaveNetModel (neural_speech/models/wavenet.py: biases, scalar input, global / local conditioning).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic lives in tensorflow-gpu==1.7.0 (tf.nn.conv1d, softmax_cross_entropy_with_logits),
which cannot be imported here, and the reference holds no fixtures for this path.  Restated in PyTorch-CPU float64
with autograd, each block citing the reference lines; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import it.

Shapes follow the reference: audio [N, T] -> mu-law ids [N, T]; the network sees ids[:, :T-1] one-hot; every VALID
causal convolution of dilation d shortens the series by d; the loss compares the last T - receptive_field logits
with ids[:, receptive_field:].  `network_full` restates models/wavenet.py with its options; with all of them off it is
`network` (asserted in tests/test_wavenet_options_gpu.py)."""
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


# ---------------------------------------------------------------------------------------------------------------------
# The full WaveNetModel (neural_speech/models/wavenet.py).  Options: use_biases (:217-232, :249-254, :342-346, :362-366,
# :473-479), scalar_input + initial_filter_width (:162-173, :679-682), global conditioning by category or by vector
# (:144-159, :206-215, :302-322, :573-608), local conditioning (:217-226, :324-340).
def receptive_field_full(hp):
    """wavenet.py:127-134."""
    fw = hp["filter_width"]
    rf = (fw - 1) * sum(dilations(hp)) + 1
    return rf + (hp["initial_filter_width"] - 1 if hp["scalar_input"] else fw - 1)


def embed_gc(p, hp, global_condition):
    """wavenet.py:573-608: [N] category ids -> rows of gc_embedding, or [N, gc_channels] vectors as they are; -> [N, 1, gc]."""
    if global_condition is None:
        return None
    if hp.get("gc_category_cardinality"):
        e = p["wavenet/embeddings/gc_embedding"][torch.as_tensor(global_condition).long()]
    else:
        e = torch.as_tensor(global_condition, dtype=p["wavenet/causal_layer/filter"].dtype)
        assert e.shape[-1] == hp["gc_channels"]
    return e.reshape(e.shape[0], 1, hp["gc_channels"])


def network_full(p, hp, x, gc=None, lc=None):
    """wavenet.py:439-485 (_create_network with _create_dilation_layer :264-380).  x [N, T0, Q] one-hot or [N, T0, 1]
    scalar; gc [N, 1, gc_channels]; lc [N, 1 or the layer's length, lc_channels] - the reference adds the 1x1
    convolution of the condition to the dilated convolution's output as it is (:324-340), which TensorFlow accepts
    when the two time axes agree or the condition's is 1; a condition given on the network input's T0 positions is
    taken at the positions the layer's outputs stand for (its last T rows), the only reading under which a
    per-sample condition conditions the sample it belongs to."""
    bias = bool(hp["use_biases"])
    cur = causal_conv(x, p["wavenet/causal_layer/filter"], 1)                                          # :256-262
    out_w = x.shape[1] - receptive_field_full(hp) + 1
    skips = 0
    for i, d in enumerate(dilations(hp)):
        pre = "wavenet/dilated_stack/layer%d/" % i
        f = causal_conv(cur, p[pre + "filter"], d)                                                     # :294-295
        g = causal_conv(cur, p[pre + "gate"], d)
        if gc is not None:                                                                             # :302-322
            f = f + gc @ p[pre + "gc_filter"][0]
            g = g + gc @ p[pre + "gc_gate"][0]
        if lc is not None:                                                                             # :324-340
            c = lc if lc.shape[1] == 1 else lc[:, lc.shape[1] - f.shape[1]:, :]
            f = f + c @ p[pre + "lc_filter"][0]
            g = g + c @ p[pre + "lc_gate"][0]
        if bias:                                                                                       # :342-346
            f = f + p[pre + "filter_bias"]
            g = g + p[pre + "gate_bias"]
        out = torch.tanh(f) * torch.sigmoid(g)                                                         # :348
        transformed = out @ p[pre + "dense"][0]                                                        # :351-353
        skip = out[:, out.shape[1] - out_w:, :] @ p[pre + "skip"][0]                                   # :356-360
        if bias:                                                                                       # :362-366
            transformed = transformed + p[pre + "dense_bias"]
            skip = skip + p[pre + "slip_bias"]                  # (sic: the variable is created as 'slip_bias', :230)
        skips = skips + skip
        cur = cur[:, cur.shape[1] - transformed.shape[1]:, :] + transformed                            # :377-380
    c1 = torch.relu(skips) @ p["wavenet/postprocessing/postprocess1"][0]                               # :468-476
    if bias:
        c1 = c1 + p["wavenet/postprocessing/postprocess1_bias"]
    c2 = torch.relu(c1) @ p["wavenet/postprocessing/postprocess2"][0]
    if bias:
        c2 = c2 + p["wavenet/postprocessing/postprocess2_bias"]
    return c2


def net_input(p, hp, audio, ids):
    """wavenet.py:667-688: the mu-law codes one-hot, or with scalar_input the waveform itself; the last sample cut."""
    dt = p["wavenet/causal_layer/filter"].dtype
    if hp["scalar_input"]:
        return torch.as_tensor(audio, dtype=dt)[:, :-1, None]
    return torch.nn.functional.one_hot(ids[:, :-1].long(), hp["quantization_channels"]).to(dt)


def loss_full(p, hp, audio, global_condition=None, local_condition=None):
    """initialize + add_loss (wavenet.py:659-725) on float audio [N, T]."""
    ids = torch.tensor(mu_law_encode(audio, hp["quantization_channels"]))
    lc = None if local_condition is None else torch.as_tensor(local_condition, dtype=p["wavenet/causal_layer/filter"].dtype)
    logits = network_full(p, hp, net_input(p, hp, audio, ids), embed_gc(p, hp, global_condition), lc)
    target = ids[:, receptive_field_full(hp):].long()
    return torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), target.reshape(-1)), logits


def predict_proba_full(p, hp, waveform, global_condition=None):
    """wavenet.py:610-632: waveform = mu-law codes [T] (one-hot input) or samples [T] (scalar input), batch 1."""
    dt = p["wavenet/causal_layer/filter"].dtype
    w = torch.as_tensor(waveform)
    x = w.to(dt)[None, :, None] if hp["scalar_input"] else torch.nn.functional.one_hot(w.long(), hp["quantization_channels"]).to(dt)[None]
    gc = None if global_condition is None else embed_gc(p, hp, np.asarray(global_condition)[None])
    return torch.softmax(network_full(p, hp, x, gc, None)[0, -1].double(), dim=0)


def generate_full(p, hp, seed_ids, uniforms, global_condition=None):
    """Sliding-window generation with the full network (what predict_proba_incremental, wavenet.py:634-657, equals once
    its queues are primed: one-hot input, filter width 2)."""
    rf = receptive_field_full(hp)
    wave = list(int(v) for v in seed_ids)
    assert len(wave) >= rf, "seed shorter than the receptive field"
    for u in uniforms:
        pr = predict_proba_full(p, hp, torch.tensor(wave[-rf:]), global_condition).numpy()
        c = np.cumsum(pr)
        wave.append(int(min(np.searchsorted(c, u * c[-1], side="right"), len(pr) - 1)))
    return np.asarray(wave, np.int32)
