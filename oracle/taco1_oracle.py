"""CPU oracle for the Tacotron-1 path (neural_speech/models/tacotron.py).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (same status as taco2_oracle.py): the arithmetic lives in tensorflow-gpu==1.7.0.
Restated in PyTorch-CPU float64, each block citing the reference call site.  GRUCell follows
tf.contrib.rnn.GRUCell: [r,u] = sigmoid([x,h].Wg + bg), c = tanh([x, r*h].Wc + bc),
h' = u*h + (1-u)*c (gate bias initialised to 1.0 - an initial VALUE, not a compute-time constant).
"""
import torch

from . import taco2_oracle as _t2
from .taco2_oracle import conv1d_bn, prenet


def softsign_dense(e, p, scope):
    """tf.layers.dense(speaker_embd, units, activation=tf.nn.softsign): x / (1 + |x|) of e . W + b."""
    z = e @ p[scope + "/kernel"] + p[scope + "/bias"]
    return z / (1.0 + z.abs())


def gru_cell(x, h, Wg, bg, Wc, bc):
    ru = torch.sigmoid(torch.cat([x, h], -1) @ Wg + bg)
    r, u = ru.chunk(2, dim=-1)
    c = torch.tanh(torch.cat([x, r * h], -1) @ Wc + bc)
    return u * h + (1 - u) * c


def bigru(x, lengths, p, scope, units, h0=None):
    """modules.py:172-181: bidirectional_dynamic_rnn(GRUCell, GRUCell, initial_state_fw = initial_state_bw = s,
    sequence_length=lengths): past a sequence's length the state is carried unchanged and the output is zero, so the
    backward cell meets the initial state at the sequence's last valid step."""
    N, T, _ = x.shape
    if lengths is None:
        lengths = torch.full((N,), T, dtype=torch.long)
    outs = []
    for d in ("fw", "bw"):
        pre = "%s/%s/gru_cell" % (scope, d)
        h = x.new_zeros(N, units) if h0 is None else h0
        ys = [None] * T
        order = range(T) if d == "fw" else range(T - 1, -1, -1)
        for t in order:
            h2 = gru_cell(x[:, t], h, p[pre + "/gates/kernel"], p[pre + "/gates/bias"],
                          p[pre + "/candidate/kernel"], p[pre + "/candidate/bias"])
            m = (t < lengths).to(x.dtype)[:, None]
            h = m * h2 + (1 - m) * h
            ys[t] = m * h2
        outs.append(torch.stack(ys, dim=1))
    return torch.cat(outs, dim=2)


def highwaynet(x, p, scope):
    """modules.py:185-191: H = relu(dense), T = sigmoid(dense, bias init -1): H*T + x*(1-T)."""
    # _t2._relu = torch.relu unless a test records / forces the branch masks (taco2_oracle.MASK_LOG / MASK_FORCE)
    h = _t2._relu(x @ p[scope + "/H/kernel"] + p[scope + "/H/bias"])
    t = torch.sigmoid(x @ p[scope + "/T/kernel"] + p[scope + "/T/bias"])
    return h * t + x * (1.0 - t)


def cbhg(x, lengths, p, scope, K, c, training, bn_updates, num_highways=4, gru_units=128, speaker_embd=None):
    """modules.py:133-182.  The max-pool result is overwritten before use: the first
    projection reads the conv bank directly.  With a speaker embedding (modules.py:157-169) every highway layer reads
    [h | softsign(dense(e)) tiled over time] - the projection as wide as h, so the width doubles per layer - and a
    further projection is the initial state of both GRU directions."""
    bank = torch.cat([conv1d_bn(x, p, "%s/conv_bank/conv1d_%d" % (scope, k), torch.relu, training, bn_updates)
                      for k in range(1, K + 1)], dim=-1)
    y = bank
    for i, size in enumerate(c[:-1]):
        y = conv1d_bn(bank, p, "%s/proj_%d" % (scope, i + 1), torch.relu, training, bn_updates)
    y = conv1d_bn(y, p, "%s/proj_%d" % (scope, len(c)), None, training, bn_updates)
    hw = y + x
    if hw.shape[2] != 128:
        hw = hw @ p[scope + "/dense/kernel"] + p[scope + "/dense/bias"]
    for i in range(num_highways):
        if speaker_embd is not None:
            sp = softsign_dense(speaker_embd, p, "%s/highway_%d/dense" % (scope, i))       # [N, width of hw]
            hw = torch.cat([hw, sp[:, None, :].expand(-1, hw.shape[1], -1)], dim=-1)
        hw = highwaynet(hw, p, "%s/highway_%d/highway" % (scope, i))
    h0 = softsign_dense(speaker_embd, p, scope + "/dense") if speaker_embd is not None else None
    return bigru(hw, lengths, p, scope + "/bidirectional_rnn", gru_units, h0)


def bahdanau_alignments(query, keys, lengths, p, scope):
    """tf.contrib.seq2seq.BahdanauAttention (modules.py:76-82): score = sum v*tanh(keys + Wq.query),
    -inf past the memory length, softmax."""
    q = query @ p[scope + "/query_layer/kernel"]
    score = (p[scope + "/attention_v"] * torch.tanh(keys + q[:, None, :])).sum(dim=2)
    T = keys.shape[1]
    mask = torch.arange(T)[None, :] < lengths[:, None]
    score = torch.where(mask, score, torch.full_like(score, -float("inf")))
    return torch.softmax(score, dim=1)


def taco1_forward(p, hp, inputs, input_lengths, mel_targets=None, linear_targets=None, max_iters=None, speaker_ids=None):
    """tacotron.py:16-122.  Training iff linear_targets is given.  num_speakers > 1 (tacotron.py:41-48): the embedding
    of speaker_ids goes to the encoder CBHG (tacotron.py:57-62) and to the decoder's PrenetWrapper (modules.py:95-97,
    rnn_wrappers.py:28-30); the post CBHG gets none (tacotron.py:93)."""
    training = linear_targets is not None
    N, Ti = inputs.shape
    M, r = hp["num_mels"], hp["outputs_per_step"]
    bn_updates = {}
    lengths = input_lengths.long()
    x = p["embedding/embedding"][inputs.long()]
    x = prenet(x, p, "prenet")                                                    # tacotron.py:52-56
    spk = p["speaker/speaker_embed"][speaker_ids.long()] if hp.get("num_speakers", 1) > 1 else None
    enc = cbhg(x, lengths, p, "encoder_cbhg", hp["encoder_cbhg_banks"], hp["encoder_cbhg_bank_sizes"], training,
               bn_updates, speaker_embd=spk)                                      # [N,Ti,256]
    spk_dec = softsign_dense(spk, p, "decoder/dense") if spk is not None else None
    mask = (torch.arange(Ti)[None, :] < lengths[:, None]).to(enc.dtype)
    values = enc * mask[:, :, None]
    keys = values @ p["attention_decoder/memory_layer/kernel"]
    A = hp["attention_dim"]
    D = hp["decoder_dim"]
    h_att = enc.new_zeros(N, A)
    h1 = enc.new_zeros(N, D)
    h2 = enc.new_zeros(N, D)
    ctx = enc.new_zeros(N, values.shape[2])
    if training:
        fed = mel_targets[:, r - 1::r, :]
        steps = fed.shape[1]
        if max_iters is not None:
            steps = min(steps, max_iters)
    else:
        steps = max_iters if max_iters is not None else hp["max_iters"]
    frame = enc.new_zeros(N, M)
    outs, aligns = [], []
    Dd = "decoder"
    for s in range(steps):
        pre = prenet(torch.cat([frame, ctx], -1), p, Dd + "/decoder_prenet")      # Q8
        if spk_dec is not None:
            pre = torch.cat([pre, spk_dec], -1)                                   # rnn_wrappers.py:28-30
        h_att = gru_cell(pre, h_att, p[Dd + "/attention_gru/gates/kernel"], p[Dd + "/attention_gru/gates/bias"],
                         p[Dd + "/attention_gru/candidate/kernel"], p[Dd + "/attention_gru/candidate/bias"])
        align = bahdanau_alignments(h_att, keys, lengths, p, Dd + "/attention")
        ctx = (align[:, :, None] * values).sum(dim=1)
        x1 = torch.cat([h_att, ctx], -1) @ p[Dd + "/attention_projection/kernel"] + p[Dd + "/attention_projection/bias"]
        h1 = gru_cell(x1, h1, p[Dd + "/gru_1/gates/kernel"], p[Dd + "/gru_1/gates/bias"],
                      p[Dd + "/gru_1/candidate/kernel"], p[Dd + "/gru_1/candidate/bias"])
        y1 = x1 + h1                                                              # ResidualWrapper
        h2 = gru_cell(y1, h2, p[Dd + "/gru_2/gates/kernel"], p[Dd + "/gru_2/gates/bias"],
                      p[Dd + "/gru_2/candidate/kernel"], p[Dd + "/gru_2/candidate/bias"])
        y2 = y1 + h2
        out = y2 @ p[Dd + "/output_projection/kernel"] + p[Dd + "/output_projection/bias"]
        outs.append(out)
        aligns.append(align)
        frame = fed[:, s, :] if training else out[:, -M:]
    mel_outputs = torch.stack(outs, dim=1).reshape(N, -1, M)                      # tacotron.py:89
    post = cbhg(mel_outputs, None, p, "post_cbhg", hp["post_cbhg_banks"], list(hp["post_cbhg_bank_sizes"]) + [M],
                training, bn_updates)
    linear_outputs = post @ p["dense/kernel"] + p["dense/bias"]
    return dict(mel_outputs=mel_outputs, linear_outputs=linear_outputs, alignments=torch.stack(aligns, dim=2),
                bn_updates=bn_updates, encoder_outputs=enc)


def taco1_loss(hp, out, mel_targets, linear_targets):
    """tacotron.py:124-133 (priority band 3000 Hz -> 307 bins at the shipped config)."""
    mel_loss = (mel_targets - out["mel_outputs"]).abs().mean()
    l1 = (linear_targets - out["linear_outputs"]).abs()
    n_priority = int(3000 / (hp["sample_rate"] * 0.5) * hp["num_freq"])
    linear_loss = 0.5 * l1.mean() + 0.5 * l1[:, :, :n_priority].mean()
    return mel_loss + linear_loss, mel_loss, linear_loss


def learning_rate(hp, step):
    """tacotron.py:143-146,186-190: Noam schedule iff decay_learning_rate."""
    if not hp["decay_learning_rate"]:
        return hp["initial_learning_rate"]
    warm = 4000.0
    s = float(step + 1)
    return hp["initial_learning_rate"] * warm ** 0.5 * min(s * warm ** -1.5, s ** -0.5)
