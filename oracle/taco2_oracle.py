"""CPU oracle for the Tacotron-2 path of MLCogUP/nspeech.  TEST INFRASTRUCTURE ONLY.

This file is a restatement, in plain PyTorch-CPU tensor ops (float64 by default), of what the
reference's TensorFlow-1.7 graph computes.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it; the product path (nspeech_amd/) never does.

PARITY UNPINNED: the arithmetic lives in tensorflow-gpu==1.7.0 (requirements.txt:14), which is
not under /root/reference and not installable here, and the reference ships no tests, golden
vectors or checkpoints for this path.  Semantics of the TF ops are restated
from their published behaviour; every block cites the reference call site it follows.  Each block
is cross-checked in tests/test_oracle.py against an independent formulation (torch.nn.LSTM after
gate re-ordering, F.conv1d / F.batch_norm, torch.optim.Adam).

Weight layouts are TensorFlow's: conv kernel [k, C_in, C_out], dense [in, out],
LSTM kernel [(in+h), 4h] with gate order i, j, f, o and forget_bias 1.0 added at compute time.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3          # tf.layers.batch_normalization default epsilon
BN_MOMENTUM = 0.99     # tf.layers.batch_normalization default momentum

# Test aid: when set to a list, every ReLU appends its branch mask (pre-activation > 0) in call order.  A float32
# implementation whose forward pass differs by rounding can land on the other side of a kink whose pre-activation is
# within that rounding of zero; the parity tests compare gradients on inputs where both take the same branch everywhere.
MASK_LOG = None
# Test aid: when set to a list of masks (same call order), every ReLU takes ITS branch from that list instead of from the
# sign of its own pre-activation: relu(x) := x * mask.  With the masks recorded by the implementation under test the
# oracle differentiates the same piecewise-linear branch that implementation took - the two functions then differ only
# by arithmetic, and every gradient can be compared in the max norm at sizes where some of the ~1e6 pre-activations
# always lie within rounding of zero (the outputs move by |pre-activation| <~ 1e-5 at the flipped elements, no more).
MASK_FORCE = None
# Test aid, read with MASK_FORCE: when set to a list, every forced ReLU appends (number of elements whose own sign
# differs from the forced branch, the largest |pre-activation| among them, the site's rms pre-activation, its element
# count).  The tests bound both: a forced branch may differ from the oracle's own only where the pre-activation lies
# within the implementation's rounding of zero - a wrong epilogue (ReLU, gate, row mask) of the implementation shows up
# here as flips at pre-activations of ordinary size instead of being copied into the oracle.
FLIP_LOG = None
# Test aid: the `zoneout` argument taco2_forward() uses when its caller passes none (tests/test_zoneout_gpu.py sets it
# around the shared oracle drivers of tests/util.py).  None = the reference's plain decoder cells.
ZONEOUT = None
# LSTMBlockCell's cell_clip for every cell below; taco2_forward() sets it from hp["lstm_cell_clip"] (0 / absent = None = off)
CELL_CLIP = None


def _relu(x):
    if MASK_LOG is not None:
        MASK_LOG.append((x.detach() > 0).numpy())
    if MASK_FORCE is not None:
        mb = torch.from_numpy(MASK_FORCE.pop(0))
        assert mb.shape == x.shape, (mb.shape, x.shape)
        if FLIP_LOG is not None:
            xd = x.detach()
            fl = (xd > 0) != mb.bool()
            FLIP_LOG.append((int(fl.sum()), float(xd[fl].abs().max()) if bool(fl.any()) else 0.0,
                             float(xd.pow(2).mean().sqrt()), xd.numel()))
        return x * mb.to(x.dtype)
    return torch.relu(x)


# ------------------------------------------------------------------ building blocks
def conv1d_bn(x, p, scope, activation, training, bn_updates=None):
    """modules.py:194-198: tf.layers.conv1d(padding='same', activation) THEN batch_normalization.

    x [N,T,C_in].  'same' with stride 1 pads (k-1)//2 on the left and the rest on the right.
    Statistics run over every (n,t) position, padding frames included (no masking anywhere)."""
    W = p[scope + "/conv1d/kernel"]
    b = p[scope + "/conv1d/bias"]
    k = W.shape[0]
    padl = (k - 1) // 2
    padr = k - 1 - padl
    xp = F.pad(x.transpose(1, 2), (padl, padr))
    y = F.conv1d(xp, W.permute(2, 1, 0)) + b[None, :, None]
    y = y.transpose(1, 2)
    if activation is not None:
        y = _relu(y) if activation is torch.relu else activation(y)
    g = p[scope + "/batch_normalization/gamma"]
    be = p[scope + "/batch_normalization/beta"]
    if training:
        mean = y.mean(dim=(0, 1))
        var = y.var(dim=(0, 1), unbiased=False)
        if bn_updates is not None:
            mm = p[scope + "/batch_normalization/moving_mean"]
            mv = p[scope + "/batch_normalization/moving_variance"]
            bn_updates[scope + "/batch_normalization/moving_mean"] = (
                mm * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM))
            bn_updates[scope + "/batch_normalization/moving_variance"] = (
                mv * BN_MOMENTUM + var.detach() * (1 - BN_MOMENTUM))
    else:
        mean = p[scope + "/batch_normalization/moving_mean"]
        var = p[scope + "/batch_normalization/moving_variance"]
    return g * (y - mean) / torch.sqrt(var + BN_EPS) + be


def lstm_block_cell(x, c, h, kernel, bias):
    """tf.contrib.rnn.LSTMBlockCell: [i,j,f,o] = [x,h].W + b; forget_bias = 1.0; no peepholes.

    [3P] assumption, stated because nothing in this container can check it (TensorFlow 1.7 is absent): the cell
    state is NOT clipped.  The reference constructs every cell as `LSTMBlockCell(n)` with no further argument
    (modules.py:41-42,90; tacotron2.py:69-70), so TF 1.7's default applies.  The fused LSTMBlockCell op carries a
    `cell_clip` attribute; as far as the author of this restatement recalls the 1.5+ Python wrapper
    (contrib/rnn/python/ops/lstm_ops.py: `cell_clip=None` -> the op is handed -1, "no clipping"; the releases before
    it had `clip_cell=True` and the op's own default of 3), 1.7 does not clip - SURVEY Appendix C says the same - but
    that is memory, not a checked fact.  It matters: with random-initialised weights the expand BiLSTM's cell state
    reaches |c| = 3.4 over the 1000 steps of the benchmark shape (measured, tests/test_taco2_fullwidth_gpu.py prints it),
    so a build that clips at 3 would differ from this restatement - and from the kernels, which follow it - there."""
    z = torch.cat([x, h], dim=-1) @ kernel + bias
    i, j, f, o = z.chunk(4, dim=-1)
    c2 = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
    if CELL_CLIP:
        # the fused op clips cs in the forward pass (lstm_ops: cs = cs.cwiseMin(cell_clip).cwiseMax(-cell_clip)) and its
        # gradient kernel (LSTMBlockCellBprop) has no term for it: clip with a straight-through gradient  [3P, recalled]
        c2 = c2 + (c2.clamp(-CELL_CLIP, CELL_CLIP) - c2).detach()
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return c2, h2


def zoneout_masks(seed, thr, steps, N, H):
    """The counter-based keep masks of the kernels under test (include/nspeech_hip.h, ns_lstm_seq_params; csrc/common.h
    ns_zone_keep), restated in NumPy integer arithmetic: mask[t, n, u] = (mix(seed, t, n, u) >> 8) < thr, mix = three
    rounds of the murmur3 32-bit finaliser.  True = the unit keeps its old value.  Bit-exact by construction (uint32
    wrap-around on both sides); tests/test_zoneout_gpu.py pins it against the kernel's own output."""
    import numpy as np

    def fmix(x):
        x = x.astype(np.uint32)
        x ^= x >> np.uint32(16)
        x = (x.astype(np.uint64) * np.uint64(0x85EBCA6B)).astype(np.uint32)
        x ^= x >> np.uint32(13)
        x = (x.astype(np.uint64) * np.uint64(0xC2B2AE35)).astype(np.uint32)
        x ^= x >> np.uint32(16)
        return x

    def mul(a, c):
        return (a.astype(np.uint64) * np.uint64(c)).astype(np.uint32)

    t = np.arange(steps, dtype=np.uint32)[:, None, None]
    n = np.arange(N, dtype=np.uint32)[None, :, None]
    u = np.arange(H, dtype=np.uint32)[None, None, :]
    x = fmix(np.uint32(seed) ^ mul(t, 0x9E3779B9))
    x = fmix(x ^ mul(n, 0x7FEB352D))
    x = fmix(x ^ mul(u, 0x846CA68B))
    return (x >> np.uint32(8)) < np.uint32(thr)


def zoneout_cell(x, c, h, kernel, bias, keep_c=None, keep_h=None, rate=None):
    """Zoneout LSTM (Krueger et al. 2017) around lstm_block_cell: the plain cell proposes (c', h'), each unit keeps its
    old value where its mask is set (training: keep_c / keep_h, bool [N, H]) or, at inference, moves to the expectation
    rate * old + (1 - rate) * new.  The zoned h is both the cell's output and its recurrent state.  Not in the
    reference (plain cells, tacotron2.py:69-70); rate 0 / no masks = lstm_block_cell."""
    c2, h2 = lstm_block_cell(x, c, h, kernel, bias)
    if keep_c is not None:
        kc = torch.from_numpy(keep_c).to(c2.dtype)
        kh = torch.from_numpy(keep_h).to(c2.dtype)
        return kc * c + (1 - kc) * c2, kh * h + (1 - kh) * h2
    if rate:
        return rate * c + (1 - rate) * c2, rate * h + (1 - rate) * h2
    return c2, h2


def bilstm(x, lengths, p, scope, units):
    """modules.py:40-49: tf.nn.bidirectional_dynamic_rnn(LSTMBlockCell, LSTMBlockCell,
    sequence_length=lengths).  Past its length an example emits zeros and freezes its state;
    the backward cell sees each sequence reversed over its own length."""
    N, T, _ = x.shape
    if lengths is None:
        lengths = torch.full((N,), T, dtype=torch.long)
    outs = []
    for d in ("fw", "bw"):
        K = p["%s/%s/lstm_cell/kernel" % (scope, d)]
        b = p["%s/%s/lstm_cell/bias" % (scope, d)]
        c = x.new_zeros(N, units)
        h = x.new_zeros(N, units)
        ys = [None] * T
        order = range(T) if d == "fw" else range(T - 1, -1, -1)
        for t in order:
            c2, h2 = lstm_block_cell(x[:, t], c, h, K, b)
            m = (t < lengths).to(x.dtype)[:, None]
            c = m * c2 + (1 - m) * c
            h = m * h2 + (1 - m) * h
            ys[t] = m * h2
        outs.append(torch.stack(ys, dim=1))
    return torch.cat(outs, dim=2)


def conv_and_lstm(x, lengths, p, scope, layers, units, training, bn_updates):
    """modules.py:30-49."""
    for i in range(layers):
        act = torch.relu if i < layers - 1 else None
        x = conv1d_bn(x, p, "%s/conv_%d" % (scope, i), act, training, bn_updates)
    return bilstm(x, lengths, p, scope + "/encoder_lstm", units)


def postnet(x, p, scope, layers, training, bn_updates):
    """modules.py:52-58: 5 x conv1d (tanh on all but the last) then Dense back to num_mels."""
    inp = x
    for i in range(layers):
        act = torch.tanh if i < layers - 1 else None
        x = conv1d_bn(x, p, "%s/postnet_conv_%d" % (scope, i), act, training, bn_updates)
    return x @ p[scope + "/dense/kernel"] + p[scope + "/dense/bias"]


def prenet(x, p, scope):
    """modules.py:21-27 via rnn_wrappers.py:25-27.  tf.layers.dropout is called without
    training=True, so it is the identity."""
    x = _relu(x @ p[scope + "/dense_1/kernel"] + p[scope + "/dense_1/bias"])
    x = _relu(x @ p[scope + "/dense_2/kernel"] + p[scope + "/dense_2/bias"])
    return x


def location_sensitive_alignments(query, prev_align, keys, lengths, p, scope):
    """attention.py:30-60 (+ BahdanauAttention's masked softmax, score_mask_value = -inf)."""
    N, T = prev_align.shape
    Wc = p[scope + "/location_conv/kernel"]           # [7,1,20], 'same', no bias
    k = Wc.shape[0]
    padl = (k - 1) // 2
    xp = F.pad(prev_align[:, None, :], (padl, k - 1 - padl))
    f = F.conv1d(xp, Wc.permute(2, 1, 0)).transpose(1, 2)            # [N,T,20]
    loc = f @ p[scope + "/location_layer/kernel"]                     # [N,T,256]
    q = query @ p[scope + "/query_layer/kernel"]                      # [N,256]
    v = p[scope + "/attention_v"]
    score = (v * torch.tanh(keys + q[:, None, :] + loc)).sum(dim=2)   # [N,T]
    mask = torch.arange(T)[None, :] < lengths[:, None]
    score = torch.where(mask, score, torch.full_like(score, -float("inf")))
    return torch.softmax(score, dim=1)


# ------------------------------------------------------------------ the model
def speaker_projection(p, speaker_ids, scope):
    """tacotron2.py:40-47 (lookup in speaker/speaker_embed when num_speakers > 1) and rnn_wrappers.py:28-29
    (tf.layers.dense(speaker_embd, 128, activation=softsign), softsign(x) = x / (1 + |x|))."""
    e = p["speaker/speaker_embed"][speaker_ids.long()]
    return F.softsign(e @ p[scope + "/dense/kernel"] + p[scope + "/dense/bias"])


def taco2_forward(p, hp, inputs, input_lengths, mel_targets=None, linear_targets=None,
                  max_iters=None, collect=False, speaker_ids=None, zoneout=None):
    """tacotron2.py:15-128.  Training mode iff linear_targets is given (line 34).

    zoneout: None (the reference: plain decoder cells) or, for the zoneout option of the build (hparam zoneout_rate),
    dict(rate=r, masks={1: (keep_c, keep_h), 2: (...)}) with bool arrays [steps, N, units] for a training pass (see
    zoneout_masks) or dict(rate=r) for the inference expectation.
    p: dict name -> torch tensor (TF layouts, names below 'model/inference/').
    Returns dict with mel_outputs, linear_outputs, alignments [N,T_in,steps], decoder_outputs
    and bn_updates (the UPDATE_OPS moving-average assignments, tacotron2.py:157-161)."""
    training = linear_targets is not None
    global CELL_CLIP
    CELL_CLIP = float(hp.get("lstm_cell_clip", 0.0) or 0.0) or None      # read by lstm_block_cell for every cell of this pass
    if zoneout is None:
        zoneout = ZONEOUT
    N, Ti = inputs.shape
    M = hp["num_mels"]
    r = hp["outputs_per_step"]
    bn_updates = {}
    lengths = input_lengths.long()

    x = p["embedding/embedding"][inputs.long()]                                   # modules.py:8-18
    enc = conv_and_lstm(x, lengths, p, "encoder", hp["encoder_conv_layers"],
                        hp["encoder_lstm_units"], training, bn_updates)           # [N,Ti,512]

    # BahdanauAttention.__init__: values = memory zeroed past its length; keys = memory_layer(values)
    mask = (torch.arange(Ti)[None, :] < lengths[:, None]).to(enc.dtype)
    values = enc * mask[:, :, None]
    D = "decoder"
    keys = values @ p["attention_decoder/memory_layer/kernel"]

    att_units = hp["attention_dim"]
    dec_units = hp["decoder_lstm_units"]
    c_att = enc.new_zeros(N, att_units); h_att = enc.new_zeros(N, att_units)
    c1 = enc.new_zeros(N, dec_units); h1 = enc.new_zeros(N, dec_units)
    c2 = enc.new_zeros(N, dec_units); h2 = enc.new_zeros(N, dec_units)
    ctx = enc.new_zeros(N, values.shape[2])
    align = enc.new_zeros(N, Ti)

    if training:
        fed = mel_targets[:, r - 1::r, :]                       # helpers.py:49-53
        steps = fed.shape[1]
        if max_iters is not None:
            steps = min(steps, max_iters)
    else:
        steps = max_iters if max_iters is not None else hp["max_iters"]

    # multi-speaker: attention_decoder() hands speaker_embd to the PrenetWrapper only (modules.py:95-97; the
    # ConcatOutputAndAttentionWrapper at :104-105 is built without it)
    spk = speaker_projection(p, speaker_ids, D) if hp.get("num_speakers", 1) > 1 else None
    frame = enc.new_zeros(N, M)                                 # <GO>, helpers.py:80-82
    outs, aligns = [], []
    trace = {"h_att": [], "ctx": [], "h1": [], "h2": []}
    for s in range(steps):
        cell_in = torch.cat([frame, ctx], dim=-1)               # AttentionWrapper cell_input_fn (Q8)
        pre = prenet(cell_in, p, D + "/decoder_prenet")
        if spk is not None:
            pre = torch.cat([pre, spk], dim=-1)                 # rnn_wrappers.py:30
        c_att, h_att = lstm_block_cell(pre, c_att, h_att, p[D + "/attention_lstm/kernel"],
                                       p[D + "/attention_lstm/bias"])
        align = location_sensitive_alignments(h_att, align, keys, lengths, p, D + "/attention")
        ctx = (align[:, :, None] * values).sum(dim=1)
        x1 = torch.cat([h_att, ctx], dim=-1)                    # rnn_wrappers.py:58-64
        if zoneout is None:
            c1, h1 = lstm_block_cell(x1, c1, h1, p[D + "/lstm_1/kernel"], p[D + "/lstm_1/bias"])
            c2, h2 = lstm_block_cell(h1, c2, h2, p[D + "/lstm_2/kernel"], p[D + "/lstm_2/bias"])
        else:
            zm = zoneout.get("masks")
            k1 = (zm[1][0][s], zm[1][1][s]) if zm else (None, None)
            k2 = (zm[2][0][s], zm[2][1][s]) if zm else (None, None)
            c1, h1 = zoneout_cell(x1, c1, h1, p[D + "/lstm_1/kernel"], p[D + "/lstm_1/bias"], k1[0], k1[1], zoneout["rate"])
            c2, h2 = zoneout_cell(h1, c2, h2, p[D + "/lstm_2/kernel"], p[D + "/lstm_2/bias"], k2[0], k2[1], zoneout["rate"])
        out = h2 @ p[D + "/output_projection/kernel"] + p[D + "/output_projection/bias"]
        outs.append(out)
        aligns.append(align)
        if collect:
            trace["h_att"].append(h_att); trace["ctx"].append(ctx)
            trace["h1"].append(h1); trace["h2"].append(h2)
        if training:
            frame = fed[:, s, :]                                # helpers.py:73-77
        else:
            frame = out[:, -M:]                                 # helpers.py:32-38 (never stops, Q7)

    decoder_outputs = torch.stack(outs, dim=1).reshape(N, -1, M)                 # tacotron2.py:86
    post = postnet(decoder_outputs, p, "decoder_postnet", hp["postnet_conv_layers"], training,
                   bn_updates)
    mel_outputs = decoder_outputs + post                                          # tacotron2.py:95
    exp = conv_and_lstm(mel_outputs, None, p, "expand", hp["expand_conv_layers"],
                        hp["expand_lstm_units"], training, bn_updates)
    linear_outputs = exp @ p["dense/kernel"] + p["dense/bias"]                    # tacotron2.py:107
    alignments = torch.stack(aligns, dim=2)                                       # [N,Ti,steps]
    res = dict(mel_outputs=mel_outputs, linear_outputs=linear_outputs, alignments=alignments,
               decoder_outputs=decoder_outputs, bn_updates=bn_updates, encoder_outputs=enc,
               keys=keys)
    if collect:
        res["trace"] = {k: torch.stack(v, dim=1) for k, v in trace.items()}
    return res


def taco2_loss(hp, out, mel_targets, linear_targets):
    """tacotron2.py:130-139 (unmasked means; priority band int(2000/(sr/2)*num_freq) bins)."""
    mel_loss = (mel_targets - out["mel_outputs"]).abs().mean()
    l1 = (linear_targets - out["linear_outputs"]).abs()
    n_priority = int(2000 / (hp["sample_rate"] * 0.5) * hp["num_freq"])
    linear_loss = 0.5 * l1.mean() + 0.5 * l1[:, :, :n_priority].mean()
    return mel_loss + linear_loss, mel_loss, linear_loss


def learning_rate(hp, step):
    """tacotron2.py:150-151: exponential_decay(lr0, step, halflife, 0.5), non-staircase."""
    return hp["initial_learning_rate"] * 0.5 ** (step / hp["learning_rate_decay_halflife"])


def clip_by_global_norm(grads, clip):
    """tf.clip_by_global_norm: g * clip / max(global_norm, clip)."""
    gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    scale = clip / max(gn, clip)
    return {k: g * scale for k, g in grads.items()}, gn


def adam_step(p, grads, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps).
    `step` is the 1-based count of this update."""
    lr_t = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    for k in grads:
        m[k] = beta1 * m[k] + (1 - beta1) * grads[k]
        v[k] = beta2 * v[k] + (1 - beta2) * grads[k] ** 2
        p[k] = p[k] - lr_t * m[k] / (torch.sqrt(v[k]) + eps)
