"""Parameter store for the Tacotron models.

All trainable tensors live in ONE flat fp32 buffer (and parallel flat buffers for the gradient,
the Adam moments and the bf16 operand shadow), addressed by TensorFlow-style scope names so
that checkpoints mirror the reference's variable names (train.py:48-49, tacotron2.py:33,
modules.py:12,33,46,54,63; deep names inside dynamic_decode are this build's own choice -
the reference ships no checkpoint to compare with).  Layouts are TensorFlow's: conv kernel
[k, C_in, C_out], dense [in, out], LSTM kernel [(in+h), 4h] with gate order i,j,f,o.

Initialisers follow the reference graph: truncated-normal(0.01) embedding (modules.py:13-17),
glorot-uniform for every other kernel (TF default), zero biases, BN gamma 1 / beta 0 /
moving mean 0 / moving variance 1.
"""
import math
from collections import OrderedDict

import numpy as np

ALIGN = 8  # elements; keeps every tensor 16-byte aligned in the bf16 shadow too


def glorot_uniform(rng, shape):
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def truncated_normal(rng, shape, std):
    x = rng.normal(0.0, std, size=shape)
    bad = np.abs(x) > 2 * std
    while bad.any():
        x[bad] = rng.normal(0.0, std, size=int(bad.sum()))
        bad = np.abs(x) > 2 * std
    return x.astype(np.float32)


class Layout(object):
    """name -> (offset, shape) inside a flat buffer."""

    def __init__(self):
        self.entries = OrderedDict()
        self.size = 0

    def add(self, name, shape):
        n = int(np.prod(shape))
        self.entries[name] = (self.size, tuple(shape))
        self.size += (n + ALIGN - 1) // ALIGN * ALIGN
        return name

    def off(self, name):
        return self.entries[name][0]

    def shape(self, name):
        return self.entries[name][1]

    def numel(self, name):
        return int(np.prod(self.entries[name][1]))


def _conv_bn(trainable, stats, scope, k, cin, cout):
    trainable.add(scope + "/conv1d/kernel", (k, cin, cout))
    trainable.add(scope + "/conv1d/bias", (cout,))
    trainable.add(scope + "/batch_normalization/gamma", (cout,))
    trainable.add(scope + "/batch_normalization/beta", (cout,))
    stats.add(scope + "/batch_normalization/moving_mean", (cout,))
    stats.add(scope + "/batch_normalization/moving_variance", (cout,))


def _lstm(trainable, scope, nin, units):
    trainable.add(scope + "/kernel", (nin + units, 4 * units))
    trainable.add(scope + "/bias", (4 * units,))


def taco2_speaker_width(hp):
    """Width of the speaker projection concatenated to the decoder prenet output when num_speakers > 1
    (rnn_wrappers.py:28-30: as wide as the prenet output, 128); 0 for a single speaker."""
    return 128 if int(getattr(hp, "num_speakers", 1) or 1) > 1 else 0


def taco2_layout(hp, vocab_size):
    """Variables of tacotron2.py:33-107 (names relative to 'model/inference/')."""
    tr, st = Layout(), Layout()
    M = hp.num_mels
    emb = hp.embedding_dim
    tr.add("embedding/embedding", (vocab_size, emb))
    n_spk = int(getattr(hp, "num_speakers", 1) or 1)
    dsp = taco2_speaker_width(hp)
    if dsp:                                       # tacotron2.py:40-47
        tr.add("speaker/speaker_embed", (n_spk, hp.speaker_embed_dim))
    cin = emb
    for i in range(hp.encoder_conv_layers):
        _conv_bn(tr, st, "encoder/conv_%d" % i, hp.encoder_conv_width, cin, hp.encoder_conv_channels)
        cin = hp.encoder_conv_channels
    for d in ("fw", "bw"):
        _lstm(tr, "encoder/encoder_lstm/%s/lstm_cell" % d, cin, hp.encoder_lstm_units)
    E = 2 * hp.encoder_lstm_units
    A = hp.attention_dim
    tr.add("attention_decoder/memory_layer/kernel", (E, A))
    tr.add("decoder/decoder_prenet/dense_1/kernel", (M + E, 256))
    tr.add("decoder/decoder_prenet/dense_1/bias", (256,))
    tr.add("decoder/decoder_prenet/dense_2/kernel", (256, 128))
    tr.add("decoder/decoder_prenet/dense_2/bias", (128,))
    if dsp:                                       # rnn_wrappers.py:28-30: dense(speaker_embd, 128, softsign)
        tr.add("decoder/dense/kernel", (hp.speaker_embed_dim, dsp))
        tr.add("decoder/dense/bias", (dsp,))
    tr.add("decoder/attention_lstm/kernel", (128 + dsp + A, 4 * A))
    tr.add("decoder/attention_lstm/bias", (4 * A,))
    tr.add("decoder/attention/query_layer/kernel", (A, A))
    tr.add("decoder/attention/location_conv/kernel", (7, 1, 20))
    tr.add("decoder/attention/location_layer/kernel", (20, A))
    tr.add("decoder/attention/attention_v", (A,))
    D = hp.decoder_lstm_units
    _lstm(tr, "decoder/lstm_1", A + E, D)
    _lstm(tr, "decoder/lstm_2", D, D)
    tr.add("decoder/output_projection/kernel", (D, M * hp.outputs_per_step))
    tr.add("decoder/output_projection/bias", (M * hp.outputs_per_step,))
    cin = M
    for i in range(hp.postnet_conv_layers):
        _conv_bn(tr, st, "decoder_postnet/postnet_conv_%d" % i, hp.postnet_conv_width, cin,
                 hp.postnet_conv_channels)
        cin = hp.postnet_conv_channels
    tr.add("decoder_postnet/dense/kernel", (cin, M))
    tr.add("decoder_postnet/dense/bias", (M,))
    cin = M
    for i in range(hp.expand_conv_layers):
        _conv_bn(tr, st, "expand/conv_%d" % i, hp.expand_conv_width, cin, hp.expand_conv_channels)
        cin = hp.expand_conv_channels
    for d in ("fw", "bw"):
        _lstm(tr, "expand/encoder_lstm/%s/lstm_cell" % d, cin, hp.expand_lstm_units)
    tr.add("dense/kernel", (2 * hp.expand_lstm_units, hp.num_freq))
    tr.add("dense/bias", (hp.num_freq,))
    return tr, st


def _gru(trainable, scope, nin, units):
    trainable.add(scope + "/gates/kernel", (nin + units, 2 * units))
    trainable.add(scope + "/gates/bias", (2 * units,))
    trainable.add(scope + "/candidate/kernel", (nin + units, units))
    trainable.add(scope + "/candidate/bias", (units,))


def cbhg_highway_widths(speaker_dim, highways=4):
    """Input width of each highway layer.  With a speaker embedding every layer first concatenates a softsign
    projection as wide as its input (modules.py:157-162), so the width doubles per layer: 256, 512, 1024, 2048."""
    return [128 * (2 ** (i + 1) if speaker_dim else 1) for i in range(highways)]


def _cbhg(tr, st, scope, K, cin, proj, gru_units=128, highways=4, speaker_dim=0):
    for k in range(1, K + 1):
        _conv_bn(tr, st, "%s/conv_bank/conv1d_%d" % (scope, k), k, cin, 128)
    c = K * 128
    for i, size in enumerate(proj):
        _conv_bn(tr, st, "%s/proj_%d" % (scope, i + 1), 3, c, size)
        c = size
    if c != 128:
        assert not speaker_dim          # the initial-state projection below would then be dense_1 in the reference
        tr.add(scope + "/dense/kernel", (c, 128))
        tr.add(scope + "/dense/bias", (128,))
    widths = cbhg_highway_widths(speaker_dim, highways)
    for i in range(highways):
        if speaker_dim:                 # modules.py:157-160: dense(speaker_embd, h.shape[-1], softsign) inside highway_i
            tr.add("%s/highway_%d/dense/kernel" % (scope, i), (speaker_dim, widths[i] // 2))
            tr.add("%s/highway_%d/dense/bias" % (scope, i), (widths[i] // 2,))
        for g in ("H", "T"):
            tr.add("%s/highway_%d/highway/%s/kernel" % (scope, i, g), (widths[i], widths[i]))
            tr.add("%s/highway_%d/highway/%s/bias" % (scope, i, g), (widths[i],))
    if speaker_dim:                     # modules.py:165-167: the BiGRU's initial state (both directions)
        tr.add(scope + "/dense/kernel", (speaker_dim, gru_units))
        tr.add(scope + "/dense/bias", (gru_units,))
    for d in ("fw", "bw"):
        _gru(tr, "%s/bidirectional_rnn/%s/gru_cell" % (scope, d), widths[-1], gru_units)


def taco1_layout(hp, vocab_size):
    """Variables of tacotron.py:36-98 (names relative to 'model/inference/')."""
    tr, st = Layout(), Layout()
    M, emb = hp.num_mels, hp.embedding_dim
    tr.add("embedding/embedding", (vocab_size, emb))
    n_spk = int(getattr(hp, "num_speakers", 1) or 1)
    dsp = taco2_speaker_width(hp)                 # the decoder prenet site is the same wrapper (rnn_wrappers.py:28-30)
    sd = hp.speaker_embed_dim if dsp else 0
    if dsp:                                       # tacotron.py:41-48
        tr.add("speaker/speaker_embed", (n_spk, sd))
    pn = list(hp.encoder_prenet)
    tr.add("prenet/dense_1/kernel", (emb, pn[0])); tr.add("prenet/dense_1/bias", (pn[0],))
    tr.add("prenet/dense_2/kernel", (pn[0], pn[1])); tr.add("prenet/dense_2/bias", (pn[1],))
    _cbhg(tr, st, "encoder_cbhg", hp.encoder_cbhg_banks, pn[1], list(hp.encoder_cbhg_bank_sizes), speaker_dim=sd)
    E, A, D = 256, hp.attention_dim, hp.decoder_dim
    tr.add("attention_decoder/memory_layer/kernel", (E, A))
    tr.add("decoder/decoder_prenet/dense_1/kernel", (M + E, 256)); tr.add("decoder/decoder_prenet/dense_1/bias", (256,))
    tr.add("decoder/decoder_prenet/dense_2/kernel", (256, 128)); tr.add("decoder/decoder_prenet/dense_2/bias", (128,))
    if dsp:
        tr.add("decoder/dense/kernel", (sd, dsp)); tr.add("decoder/dense/bias", (dsp,))
    _gru(tr, "decoder/attention_gru", 128 + dsp, A)
    tr.add("decoder/attention/query_layer/kernel", (A, A))
    tr.add("decoder/attention/attention_v", (A,))
    tr.add("decoder/attention_projection/kernel", (A + E, D)); tr.add("decoder/attention_projection/bias", (D,))
    _gru(tr, "decoder/gru_1", D, D)
    _gru(tr, "decoder/gru_2", D, D)
    tr.add("decoder/output_projection/kernel", (D, M * hp.outputs_per_step))
    tr.add("decoder/output_projection/bias", (M * hp.outputs_per_step,))
    _cbhg(tr, st, "post_cbhg", hp.post_cbhg_banks, M, list(hp.post_cbhg_bank_sizes) + [M])
    tr.add("dense/kernel", (256, hp.num_freq)); tr.add("dense/bias", (hp.num_freq,))
    return tr, st


def init_values(trainable, stats, seed=0):
    """numpy dicts (name -> array) with the reference's initial values."""
    rng = np.random.RandomState(seed)
    p, s = OrderedDict(), OrderedDict()
    for name, (_, shape) in trainable.entries.items():
        if name == "embedding/embedding":
            p[name] = truncated_normal(rng, shape, 0.01)
        elif name.endswith("/gates/bias"):          # GRUCell: gate bias starts at 1.0
            p[name] = np.ones(shape, np.float32)
        elif name.endswith("/highway/T/bias"):      # modules.py:189-190
            p[name] = np.full(shape, -1.0, np.float32)
        elif name.endswith("/bias") or name.endswith("/beta"):
            p[name] = np.zeros(shape, np.float32)
        elif name.endswith("/gamma"):
            p[name] = np.ones(shape, np.float32)
        else:
            p[name] = glorot_uniform(rng, shape)
    for name, (_, shape) in stats.entries.items():
        s[name] = (np.ones if name.endswith("moving_variance") else np.zeros)(shape, np.float32)
    return p, s
