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


def _relu(x):
    if MASK_LOG is not None:
        MASK_LOG.append((x.detach() > 0).numpy())
    if MASK_FORCE is not None:
        m = torch.from_numpy(MASK_FORCE.pop(0)).to(x.dtype)
        assert m.shape == x.shape, (m.shape, x.shape)
        return x * m
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
    """tf.contrib.rnn.LSTMBlockCell: [i,j,f,o] = [x,h].W + b; forget_bias = 1.0; no peepholes."""
    z = torch.cat([x, h], dim=-1) @ kernel + bias
    i, j, f, o = z.chunk(4, dim=-1)
    c2 = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
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
                  max_iters=None, collect=False, speaker_ids=None):
    """tacotron2.py:15-128.  Training mode iff linear_targets is given (line 34).

    p: dict name -> torch tensor (TF layouts, names below 'model/inference/').
    Returns dict with mel_outputs, linear_outputs, alignments [N,T_in,steps], decoder_outputs
    and bn_updates (the UPDATE_OPS moving-average assignments, tacotron2.py:157-161)."""
    training = linear_targets is not None
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
        c1, h1 = lstm_block_cell(x1, c1, h1, p[D + "/lstm_1/kernel"], p[D + "/lstm_1/bias"])
        c2, h2 = lstm_block_cell(h1, c2, h2, p[D + "/lstm_2/kernel"], p[D + "/lstm_2/bias"])
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
