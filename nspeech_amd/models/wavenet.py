"""simple_wavenet (neural_speech/models/wavenet_simple.py) and the full WaveNetModel (neural_speech/models/wavenet.py:
biases, scalar input, global / local conditioning - class WaveNetModel at the end) on the HIP kernels: training graph
(initialize / add_loss / add_optimizer), predict_proba and sample-by-sample generation.

MI355X-first restructuring, same results as the reference graph:
  * every series lives on ONE time grid of T0 = clip length - 1 rows per item, right-aligned: a VALID causal
    convolution of dilation d does not shorten a buffer, it moves the first valid row to the right by d
    (the reference's time_to_batch / batch_to_time reshapes, wavenet_simple.py:551-583, disappear);
  * a dilated width-2 convolution = two accumulated GEMMs over the same [N*T0, C] buffer shifted by d rows, with
    filter and gate weights side by side ([2, R, 2*Dc]) so one pair of launches feeds both halves of the gated unit;
  * the one-hot input layer (:246-252, :385-397) is a pair of table look-ups, never a [.., 256] one-hot tensor;
  * the 50 skip 1x1 convolutions and their sum (:332-336, :369) are ONE GEMM over the concatenated gated
    outputs [rows, L*Dc] x [L*Dc, S] on the rows the loss uses.
Parameters keep the reference's variable names in checkpoints (numpy_params / load_numpy_params); inside, filter
and gate of a layer share one tensor and the skip kernels are contiguous."""
import math
from collections import OrderedDict

import numpy as np
import torch

from .. import ops
from .._lib import ACT_NONE, ACT_RELU
from .params import Layout, glorot_uniform


def dilations(hp):
    return [2 ** i for _ in range(hp.dilations_depth) for i in range(hp.dilations_length)]


def receptive_field(hp, full=False):
    """wavenet_simple.py:124-128; full: wavenet.py:127-134 (a scalar input enters through initial_filter_width taps)."""
    fw = hp.filter_width
    first = hp.initial_filter_width if (full and hp.scalar_input) else fw
    return (fw - 1) * sum(dilations(hp)) + 1 + (first - 1)


def mu_law_encode(audio, q):
    """wavenet_simple.py:586-597, float32 arithmetic."""
    a = np.asarray(audio, np.float32)
    mu = np.float32(q - 1)
    mag = np.log1p(mu * np.minimum(np.abs(a), np.float32(1.0))) / np.log1p(mu)
    return ((np.sign(a) * mag + 1) / 2 * mu + np.float32(0.5)).astype(np.int32)


def mu_law_decode(ids, q):
    """wavenet_simple.py:600-608."""
    mu = q - 1
    sig = 2 * (np.asarray(ids, np.float32) / mu) - 1
    return (np.sign(sig) * (1 / mu) * ((1 + mu) ** np.abs(sig) - 1)).astype(np.float32)


class SimpleWaveNet(object):
    """create_model('simple_wavenet', hparams).  dtype: 'fp32' (exact FMA, parity tests) or 'bf16'."""

    def __init__(self, hparams, device="cuda:0", dtype="bf16", seed=0, world_size=1):
        from .. import _lib
        _lib.lib()                                  # fails loudly without the HIP library
        hp = self._hparams = hparams
        assert hp.filter_width == 2, "filter_width 2 (the reference's incremental generator has no other, wavenet.py:640-642)"
        self._options(hp)
        self.device = torch.device(device)
        self.mode = dtype
        self.T = torch.float32 if dtype == "fp32" else torch.bfloat16
        self.passes = 0
        self.world_size = world_size
        self.dil = dilations(hp)
        self.rf = receptive_field(hp, self.full)
        self.start0 = (self.IFW - 1) if self.scalar_input else 1          # first valid row of the causal layer's output
        self.L = len(self.dil)
        self.Q, self.R, self.Dc, self.S = hp.quantization_channels, hp.residual_channels, hp.dilation_channels, hp.skip_channels
        lay = self.layout = Layout()
        lay.add("causal", (self.IFW, 1, self.R) if self.scalar_input else (2, self.Q, self.R))
        for i in range(self.L):
            lay.add("fg%d" % i, (2, self.R, 2 * self.Dc))
            lay.add("dense%d" % i, (self.Dc, self.R))
        lay.add("skip", (self.L, self.Dc, self.S))
        lay.add("post1", (self.S, self.S))
        lay.add("post2", (self.S, self.Q))
        # the full model's options behind the simple model's tensors (the generator's offsets stay as they are)
        if self.gc_card:
            lay.add("gc_embedding", (self.gc_card, self.gc))
        for i in range(self.L):
            if self.gc:
                lay.add("gc%d" % i, (self.gc, 2 * self.Dc))          # [filter | gate] side by side, as fg
            if self.lc:
                lay.add("lc%d" % i, (self.lc, 2 * self.Dc))
            if self.use_biases:
                lay.add("fgb%d" % i, (2 * self.Dc,))
                lay.add("deb%d" % i, (self.R,))
        if self.use_biases:
            lay.add("skb", (self.L, self.S))
            lay.add("p1b", (self.S,))
            lay.add("p2b", (self.Q,))
        n = lay.size
        dev = self.device
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self._flat_g_full = torch.zeros((n + 3) // 4 * 4, dtype=torch.float32, device=dev)      # ns_zero clears 16-byte units
        self.flat_g = self._flat_g_full[:n]
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_s = torch.zeros(n, dtype=torch.bfloat16, device=dev) if self.T != torch.float32 else self.flat_p
        self.scal = torch.zeros(16, dtype=torch.float32, device=dev)
        self._bufs = {}
        self._sig = None
        self.global_step = 0
        self.gradient_clip = 1.0
        self.loss = None
        self.load_numpy_params(self._init_values(seed))

    full = False        # WaveNetModel below: the options of neural_speech/models/wavenet.py

    def _options(self, hp):
        """simple_wavenet builds none of them (wavenet_simple.py has no such branches)."""
        assert not hp.use_biases and not hp.scalar_input, "simple_wavenet has no biases / scalar input: --model wavenet"
        assert not (hp.gc_channels or hp.lc_channels), "simple_wavenet has no conditioning: --model wavenet"
        self.use_biases = self.scalar_input = False
        self.IFW, self.gc, self.gc_card, self.lc = 2, 0, 0, 0

    # ------------------------------------------------------------------ parameters
    def _tf_shapes(self):
        """Variable names and shapes as the reference creates them (wavenet.py:136-254; the skip bias is created under
        the name 'slip_bias', :230 - kept, it is what a checkpoint holds)."""
        d = OrderedDict()
        if self.gc_card:
            d["wavenet/embeddings/gc_embedding"] = (self.gc_card, self.gc)
        d["wavenet/causal_layer/filter"] = (self.IFW, 1, self.R) if self.scalar_input else (2, self.Q, self.R)
        for i in range(self.L):
            pre = "wavenet/dilated_stack/layer%d/" % i
            d[pre + "filter"] = (2, self.R, self.Dc)
            d[pre + "gate"] = (2, self.R, self.Dc)
            d[pre + "dense"] = (1, self.Dc, self.R)
            d[pre + "skip"] = (1, self.Dc, self.S)
            if self.gc:
                d[pre + "gc_gate"] = (1, self.gc, self.Dc)
                d[pre + "gc_filter"] = (1, self.gc, self.Dc)
            if self.lc:
                d[pre + "lc_gate"] = (1, self.lc, self.Dc)
                d[pre + "lc_filter"] = (1, self.lc, self.Dc)
            if self.use_biases:
                d[pre + "filter_bias"] = (self.Dc,)
                d[pre + "gate_bias"] = (self.Dc,)
                d[pre + "dense_bias"] = (self.R,)
                d[pre + "slip_bias"] = (self.S,)
        d["wavenet/postprocessing/postprocess1"] = (1, self.S, self.S)
        d["wavenet/postprocessing/postprocess2"] = (1, self.S, self.Q)
        if self.use_biases:
            d["wavenet/postprocessing/postprocess1_bias"] = (self.S,)
            d["wavenet/postprocessing/postprocess2_bias"] = (self.Q,)
        return d

    def _init_values(self, seed):
        """create_variable = xavier_initializer_conv2d (wavenet_simple.py:12-17): Glorot uniform per variable;
        create_bias_variable zeros (wavenet.py:29-33); a square embedding table starts as the identity (:20-26)."""
        rng = np.random.RandomState(seed)
        out = OrderedDict()
        for k, shp in self._tf_shapes().items():
            if k.endswith("_bias"):
                out[k] = np.zeros(shp, np.float32)
            elif k.endswith("gc_embedding") and shp[0] == shp[1]:
                out[k] = np.identity(shp[0], dtype=np.float32)
            else:
                out[k] = glorot_uniform(rng, shp)
        return out

    def _o(self, name):
        return self.layout.off(name)

    def load_numpy_params(self, p):
        host = np.zeros(self.layout.size, np.float32)

        def put(name, arr):
            o = self._o(name)
            host[o:o + arr.size] = np.asarray(arr, np.float32).ravel()
        put("causal", p["wavenet/causal_layer/filter"])
        skips, skb = [], []
        for i in range(self.L):
            pre = "wavenet/dilated_stack/layer%d/" % i
            put("fg%d" % i, np.concatenate([p[pre + "filter"], p[pre + "gate"]], axis=2))
            put("dense%d" % i, p[pre + "dense"][0])
            skips.append(p[pre + "skip"][0])
            if self.gc:
                put("gc%d" % i, np.concatenate([p[pre + "gc_filter"][0], p[pre + "gc_gate"][0]], axis=1))
            if self.lc:
                put("lc%d" % i, np.concatenate([p[pre + "lc_filter"][0], p[pre + "lc_gate"][0]], axis=1))
            if self.use_biases:
                put("fgb%d" % i, np.concatenate([p[pre + "filter_bias"], p[pre + "gate_bias"]]))
                put("deb%d" % i, p[pre + "dense_bias"])
                skb.append(p[pre + "slip_bias"])
        put("skip", np.stack(skips))
        put("post1", p["wavenet/postprocessing/postprocess1"][0])
        put("post2", p["wavenet/postprocessing/postprocess2"][0])
        if self.gc_card:
            put("gc_embedding", p["wavenet/embeddings/gc_embedding"])
        if self.use_biases:
            put("skb", np.stack(skb))
            put("p1b", p["wavenet/postprocessing/postprocess1_bias"])
            put("p2b", p["wavenet/postprocessing/postprocess2_bias"])
        self.flat_p.copy_(torch.from_numpy(host))
        self.refresh_shadows()

    def _unflatten(self, flat):
        host = flat.detach().float().cpu().numpy()

        def get(name):
            o = self._o(name)
            shp = self.layout.shape(name)
            return host[o:o + int(np.prod(shp))].reshape(shp).copy()
        out = OrderedDict()
        Dc = self.Dc
        if self.gc_card:
            out["wavenet/embeddings/gc_embedding"] = get("gc_embedding")
        out["wavenet/causal_layer/filter"] = get("causal")
        sk = get("skip")
        skb = get("skb") if self.use_biases else None
        for i in range(self.L):
            pre = "wavenet/dilated_stack/layer%d/" % i
            fg = get("fg%d" % i)
            out[pre + "filter"], out[pre + "gate"] = fg[:, :, :Dc].copy(), fg[:, :, Dc:].copy()
            out[pre + "dense"] = get("dense%d" % i)[None]
            out[pre + "skip"] = sk[i][None]
            for tag, n in (("gc", self.gc), ("lc", self.lc)):
                if n:
                    w = get("%s%d" % (tag, i))
                    out[pre + tag + "_gate"], out[pre + tag + "_filter"] = w[:, Dc:].copy()[None], w[:, :Dc].copy()[None]
            if self.use_biases:
                b = get("fgb%d" % i)
                out[pre + "filter_bias"], out[pre + "gate_bias"] = b[:Dc].copy(), b[Dc:].copy()
                out[pre + "dense_bias"] = get("deb%d" % i)
                out[pre + "slip_bias"] = skb[i].copy()
        out["wavenet/postprocessing/postprocess1"] = get("post1")[None]
        out["wavenet/postprocessing/postprocess2"] = get("post2")[None]
        if self.use_biases:
            out["wavenet/postprocessing/postprocess1_bias"] = get("p1b")
            out["wavenet/postprocessing/postprocess2_bias"] = get("p2b")
        return out

    def numpy_params(self):
        return self._unflatten(self.flat_p)

    def numpy_grads(self):
        return self._unflatten(self.flat_g)

    def state_dict(self):
        return dict(params=self.flat_p.cpu(), m=self.flat_m.cpu(), v=self.flat_v.cpu(), global_step=self.global_step)

    def load_state_dict(self, sd):
        self.flat_p.copy_(sd["params"]); self.flat_m.copy_(sd["m"]); self.flat_v.copy_(sd["v"])
        self.global_step = int(sd["global_step"])
        self.refresh_shadows()

    def refresh_shadows(self):
        if self.flat_s is not self.flat_p:
            n = self.layout.size
            ops.cast2d(self.flat_p, 1, n, n, self.flat_s, n, False)

    def _buf(self, name, numel, dtype):
        b = self._bufs.get(name)
        if b is None or b.numel() < numel or b.dtype != dtype:
            b = torch.zeros((numel + 7) // 8 * 8, dtype=dtype, device=self.device)       # whole 16-byte units: ops.zero
            self._bufs[name] = b
        return b

    # ------------------------------------------------------------------ network
    def _bias(self, name):
        """ns_gemm keywords for an epilogue bias (the fp32 master copy), or {}."""
        return dict(bias=self.flat_p, bias_off=self._o(name)) if self.use_biases else {}

    def _forward(self, net_in, N, T0, keep, gcrows=None, lcrows=None):
        """net_in: ids int32 [N, T0] on the device, or with scalar_input the samples [N * T0] in the compute dtype.
        gcrows / lcrows: the conditions per network-input row, [N * T0, gc] / [N * T0, lc] in the compute dtype.
        Returns (logits fp32 [N*ow, Q], ow).  keep: save what backward needs."""
        T_, W = self.T, self.flat_s
        R, Dc, S, Q, L = self.R, self.Dc, self.S, self.Q, self.L
        rows = N * T0
        ow = T0 - self.rf + 1
        assert ow >= 1, "clip shorter than the receptive field (%d samples)" % self.rf
        ops.F32_PASSES = self.passes
        xs = self._buf("xs", (L + 1) * rows * R if keep else 2 * rows * R, T_)
        outs = self._buf("outs", rows * L * Dc, T_)
        zs = self._buf("zs", (L if keep else 1) * rows * 2 * Dc, torch.float32)
        if self.scalar_input:
            # x0[m] = sum_k w[k] * s[m - (IFW-1) + k]: a product over the overlapping windows of the series (lda = 1)
            K = self.IFW
            ops.gemm(net_in, W, xs, rows - (K - 1), R, K, 1, R, R, b_mode=1, b_off=self._o("causal"), c_off=(K - 1) * R)
        else:
            ops.wavenet_input(net_in, self.flat_p, xs, N, T0, R, Q, w_off=self._o("causal"))
        start = self.start0
        for l, d in enumerate(self.dil):
            xo = (l if keep else l % 2) * rows * R
            xn = ((l + 1) if keep else (l + 1) % 2) * rows * R
            zo = (l if keep else 0) * rows * 2 * Dc
            fg = self._o("fg%d" % l)
            # z[m] = x[m-d] . W[0] + x[m] . W[1]  for rows m >= d  (+ bias, + the conditions' 1x1 convolutions)
            ops.gemm(xs, W, zs, rows - d, 2 * Dc, R, R, 2 * Dc, 2 * Dc, b_mode=1, a_off=xo, b_off=fg, c_off=zo + d * 2 * Dc,
                     **self._bias("fgb%d" % l))
            ops.gemm(xs, W, zs, rows - d, 2 * Dc, R, R, 2 * Dc, 2 * Dc, b_mode=1, a_off=xo + d * R, b_off=fg + R * 2 * Dc,
                     c_off=zo + d * 2 * Dc, accumulate=1)
            for tag, cr, Cc in (("gc", gcrows, self.gc), ("lc", lcrows, self.lc)):
                if cr is not None:
                    ops.gemm(cr, W, zs, rows - d, 2 * Dc, Cc, Cc, 2 * Dc, 2 * Dc, b_mode=1, a_off=d * Cc,
                             b_off=self._o("%s%d" % (tag, l)), c_off=zo + d * 2 * Dc, accumulate=1)
            start += d
            ops.wavenet_gate(zs[zo:], rows, Dc, T0, start, out=outs, out_off=l * Dc, ld_out=L * Dc)
            # x_next = out . dense + x
            ops.gemm(outs, W, xs, rows, R, Dc, L * Dc, R, R, b_mode=1, a_off=l * Dc, b_off=self._o("dense%d" % l), c_off=xn,
                     addend=xs, addend_off=xo, ld_add=R, **self._bias("deb%d" % l))
        # skip sum on the rows the loss uses (t >= rf - 1), relu -> post1 -> relu -> post2
        skb = {}
        if self.use_biases:          # every layer adds its skip bias: their sum rides on the one skip product
            sb = self._buf("skb_sum", S, torch.float32)
            ops.zero(sb)
            ops.colsum(self.flat_p, S, L, S, sb, x_off=self._o("skb"))
            skb = dict(bias=sb)
        t1 = self._buf("t1", N * ow * S, T_)
        for n in range(N):
            ops.gemm(outs, W, t1, ow, S, L * Dc, L * Dc, S, S, b_mode=1, a_off=(n * T0 + self.rf - 1) * L * Dc,
                     b_off=self._o("skip"), c_off=n * ow * S, act=ACT_RELU, **skb)
        c1 = self._buf("c1", N * ow * S, T_)
        ops.gemm(t1, W, c1, N * ow, S, S, S, S, S, b_mode=1, b_off=self._o("post1"), act=ACT_RELU, **self._bias("p1b"))
        logits = self._buf("logits", N * ow * Q, torch.float32)
        ops.gemm(c1, W, logits, N * ow, Q, S, S, Q, Q, b_mode=1, b_off=self._o("post2"), **self._bias("p2b"))
        return logits, ow

    def initialize(self, audio_inputs, global_conditions=None, local_conditions=None):
        """wavenet_simple.py:455-477 + add_loss :479-502: audio float [N, T] in [-1, 1]; runs the forward pass and
        the loss (its gradient wrt the logits comes out of the same kernel)."""
        audio = np.asarray(audio_inputs, np.float32)
        if audio.ndim == 1:
            audio = audio[None]
        assert self.full or (global_conditions is None and local_conditions is None), "simple_wavenet takes no conditions"
        ids = torch.from_numpy(mu_law_encode(audio, self.Q)).to(self.device)
        return self.initialize_ids(ids, audio=audio, global_conditions=global_conditions, local_conditions=local_conditions)

    def _condition_rows(self, N, T0, global_conditions, local_conditions):
        """The conditions per network-input row, in the compute dtype (wavenet.py:573-608 _embed_gc; :324-340).
        Global: [N] category ids (gc_category_cardinality) or [N, gc_channels] vectors, the same for every row of an
        item.  Local: [N, 1, lc] (every row) or [N, T0, lc] (row t's own; oracle/wavenet_oracle.py: network_full says
        how that relates to the reference's graph)."""
        dev = self.device
        gcrows = lcrows = None
        self._gc_ids = None
        if self.gc:
            assert global_conditions is not None, "gc_channels is set: initialize needs global_conditions"
            h = self._buf("gc_h", N * self.gc, torch.float32)
            if self.gc_card:
                gid = np.asarray(global_conditions, np.int64).reshape(N)
                assert gid.min() >= 0 and gid.max() < self.gc_card, "global condition outside gc_category_cardinality"
                self._gc_ids = [int(v) for v in gid]
                for n, v in enumerate(self._gc_ids):          # embedding_lookup: N rows of the table
                    ops.copy3d(self.flat_p, h, 1, 1, self.gc, (0, 0), (0, 0), src_off=self._o("gc_embedding") + v * self.gc,
                               dst_off=n * self.gc)
            else:
                g = np.asarray(global_conditions, np.float32).reshape(N, -1)
                if g.shape[1] != self.gc:
                    raise ValueError("Shape of global_condition %s does not match global_condition_channels %d."
                                     % (g.shape, self.gc))
                h[:N * self.gc].copy_(torch.from_numpy(np.ascontiguousarray(g)).to(dev).view(-1))
            gcrows = self._buf("gc_rows", N * T0 * self.gc, self.T)
            ops.copy3d(h, gcrows, N, T0, self.gc, (self.gc, 0), (T0 * self.gc, self.gc))
        else:
            assert global_conditions is None, "gc_channels is 0: no global condition expected"
        if self.lc:
            assert local_conditions is not None, "lc_channels is set: initialize needs local_conditions"
            c = np.asarray(local_conditions, np.float32)
            assert c.ndim == 3 and c.shape[0] == N and c.shape[2] == self.lc and c.shape[1] in (1, T0), \
                "local_conditions: [N, 1 or T - 1, lc_channels]"
            src = torch.from_numpy(np.ascontiguousarray(c)).to(dev).view(-1)
            lcrows = self._buf("lc_rows", N * T0 * self.lc, self.T)
            if c.shape[1] == 1:
                ops.copy3d(src, lcrows, N, T0, self.lc, (self.lc, 0), (T0 * self.lc, self.lc))
            else:
                ops.copy3d(src, lcrows, 1, N * T0, self.lc, (0, self.lc), (0, self.lc))
        else:
            assert local_conditions is None, "lc_channels is 0: no local condition expected"
        return gcrows, lcrows

    def initialize_ids(self, ids, audio=None, global_conditions=None, local_conditions=None):
        N, T = ids.shape
        T0 = T - 1
        self.ids = ids.to(self.device, torch.int32).contiguous()
        self.dims = dict(N=N, T0=T0)
        if self.scalar_input:       # the waveform itself is the network input (wavenet.py:679-682), its last sample cut
            assert audio is not None, "scalar_input: initialize(audio) (the codes do not determine the samples)"
            a = torch.from_numpy(np.ascontiguousarray(np.asarray(audio, np.float32)[:, :T0])).to(self.device)
            net_in = self._buf("scalar_in", N * T0, self.T)
            ops.copy3d(a.view(-1), net_in, 1, 1, N * T0, (0, 0), (0, 0))
        else:
            net_in = self.ids[:, :T0].contiguous()
        gcrows, lcrows = self._condition_rows(N, T0, global_conditions, local_conditions) if self.full else (None, None)
        self._cond = (gcrows, lcrows)
        logits, ow = self._forward(net_in, N, T0, keep=True, gcrows=gcrows, lcrows=lcrows)
        self.dims["ow"] = ow
        self.targets = self.ids[:, self.rf:].contiguous()                   # [N, ow]
        self.raw_output = logits[:N * ow * self.Q].view(N, ow, self.Q)
        ops.zero(self.scal)
        self.dlogits = self._buf("dlogits", N * ow * self.Q, self.T)
        ops.wavenet_softmax_ce(logits, self.Q, self.targets, N * ow, self.Q, 1.0 / (N * ow), self.scal, dlogits=self.dlogits,
                               ld_d=self.Q)
        self._net_in = net_in
        self._last = (audio, global_conditions, local_conditions)
        return self

    def add_loss(self, l2_regularization_strength=None):
        assert not l2_regularization_strength, "l2 regularisation is 0 in the shipped wavenet.yaml"
        return self

    def add_optimizer(self, global_step=0, gradient_clip=1.0):
        self.global_step = int(global_step)
        self.gradient_clip = float(gradient_clip)
        self.optimize = self.step
        return self

    def add_stats(self):
        self.stats = lambda: dict(loss=self.loss, learning_rate=self.learning_rate)
        return self

    def learning_rate_at(self, step):
        """wavenet_simple.py:513-522, 544-547: Noam schedule iff decay_learning_rate."""
        hp = self._hparams
        if not hp.decay_learning_rate:
            return hp.initial_learning_rate
        warm = 4000.0
        s = float(step + 1)
        return hp.initial_learning_rate * warm ** 0.5 * min(s * warm ** -1.5, s ** -0.5)

    # ------------------------------------------------------------------ backward
    @staticmethod
    def _sk(K):
        """split-K factor of a weight-gradient GEMM: its output tile is tiny (<= 64 x 64) and K is the row count."""
        return max(1, min(256, K // 2048))

    def backward(self):
        T_, W, g = self.T, self.flat_s, self.flat_g
        R, Dc, S, Q, L = self.R, self.Dc, self.S, self.Q, self.L
        N, T0, ow = self.dims["N"], self.dims["T0"], self.dims["ow"]
        rows, M = N * T0, N * ow
        B = self._bufs
        ops.F32_PASSES = self.passes
        ops.zero(self._flat_g_full)
        f32 = torch.float32
        t1, c1, outs, xs, zs = B["t1"], B["c1"], B["outs"], B["xs"], B["zs"]
        # post2 / post1
        ops.gemm(c1, self.dlogits, g, S, Q, M, S, Q, Q, a_mode=1, b_mode=1, c_off=self._o("post2"), accumulate=2, split_k=self._sk(M))
        if self.use_biases:
            ops.colsum(self.dlogits, Q, M, Q, g, out_off=self._o("p2b"))
        dc1 = self._buf("dc1", M * S, f32)
        ops.gemm(self.dlogits, W, dc1, M, S, Q, Q, Q, S, b_mode=0, b_off=self._o("post2"))
        dp1 = self._buf("dp1", M * S, T_)
        ops.act_bwd(dc1, c1, dp1, M, S, ACT_RELU)
        ops.gemm(t1, dp1, g, S, S, M, S, S, S, a_mode=1, b_mode=1, c_off=self._o("post1"), accumulate=2, split_k=self._sk(M))
        if self.use_biases:
            ops.colsum(dp1, S, M, S, g, out_off=self._o("p1b"))
        dt1 = self._buf("dt1", M * S, f32)
        ops.gemm(dp1, W, dt1, M, S, S, S, S, S, b_mode=0, b_off=self._o("post1"))
        dsk = self._buf("dsk", M * S, T_)
        ops.act_bwd(dt1, t1, dsk, M, S, ACT_RELU)
        if self.use_biases:          # every layer's skip bias sees the same gradient
            sb = self._buf("skb_sum", S, f32)
            ops.zero(sb)
            ops.colsum(dsk, S, M, S, sb)
            ops.copy3d(sb, g, 1, L, S, (0, 0), (0, S), dst_off=self._o("skb"), accumulate=1)
        # skip GEMM: d(outs) on the loss rows, and the stacked skip kernels
        douts = self._buf("douts", rows * L * Dc, T_)
        ops.zero(douts)
        for n in range(N):
            ao = (n * T0 + self.rf - 1) * L * Dc
            ops.gemm(dsk, W, douts, ow, L * Dc, S, S, S, L * Dc, b_mode=0, a_off=n * ow * S, b_off=self._o("skip"), c_off=ao)
            ops.gemm(outs, dsk, g, L * Dc, S, ow, L * Dc, S, S, a_mode=1, b_mode=1, a_off=ao, b_off=n * ow * S,
                     c_off=self._o("skip"), accumulate=2, split_k=self._sk(ow))
        # dilated stack, last layer first; dx of the last residual output is zero (nothing reads it)
        dx = self._buf("dx", 2 * rows * R, T_)
        ops.zero(dx)
        gcrows, lcrows = self._cond if self.full else (None, None)
        dgc = None
        if self.gc_card:             # gradient wrt the condition rows -> the embedding rows they were looked up from
            dgc = self._buf("dgc_rows", rows * self.gc, f32)
            ops.zero(dgc)
        dz = self._buf("dz", (rows + max(self.dil)) * 2 * Dc, T_)      # the tail rows are never written: zeros
        starts = [1]
        for d in self.dil:
            starts.append(starts[-1] + d)
        for l in range(L - 1, -1, -1):
            d = self.dil[l]
            cur, nxt = (l % 2) * rows * R, ((l + 1) % 2) * rows * R          # dx_l is written, dx_{l+1} is read
            fg, de = self._o("fg%d" % l), self._o("dense%d" % l)
            xo, zo = l * rows * R, l * rows * 2 * Dc
            # d(out_l) += dx_{l+1} . dense^T ;  d(dense) += out_l^T . dx_{l+1}
            dol = self._buf("dout_l", rows * Dc, T_)
            ops.gemm(dx, W, dol, rows, Dc, R, R, R, Dc, b_mode=0, a_off=nxt, b_off=de, addend=douts, addend_off=l * Dc,
                     ld_add=L * Dc)
            ops.gemm(outs, dx, g, Dc, R, rows, L * Dc, R, R, a_mode=1, b_mode=1, a_off=l * Dc, b_off=nxt, c_off=de, accumulate=2,
                     split_k=self._sk(rows))
            if self.use_biases:      # dx_{l+1} is zero on the rows layer l does not produce
                ops.colsum(dx, R, rows, R, g, x_off=nxt, out_off=self._o("deb%d" % l))
            ops.wavenet_gate(zs[zo:], rows, Dc, T0, starts[l + 1], dout=dol, ld_dout=Dc, dz=dz)
            if self.use_biases:
                ops.colsum(dz, 2 * Dc, rows, 2 * Dc, g, out_off=self._o("fgb%d" % l))
            for tag, cr, Cc in (("gc", gcrows, self.gc), ("lc", lcrows, self.lc)):
                if cr is not None:
                    ops.gemm(cr, dz, g, Cc, 2 * Dc, rows - d, Cc, 2 * Dc, 2 * Dc, a_mode=1, b_mode=1, a_off=d * Cc,
                             b_off=d * 2 * Dc, c_off=self._o("%s%d" % (tag, l)), accumulate=2, split_k=self._sk(rows))
            if dgc is not None:      # d(gc row m) += dz[m] . Wgc^T, rows m >= d
                ops.gemm(dz, W, dgc, rows - d, self.gc, 2 * Dc, 2 * Dc, 2 * Dc, self.gc, b_mode=0, a_off=d * 2 * Dc,
                         b_off=self._o("gc%d" % l), c_off=d * self.gc, accumulate=1)
            # weight gradients of the two taps (contraction over rows m >= d)
            ops.gemm(xs, dz, g, R, 2 * Dc, rows - d, R, 2 * Dc, 2 * Dc, a_mode=1, b_mode=1, a_off=xo, b_off=d * 2 * Dc, c_off=fg,
                     accumulate=2, split_k=self._sk(rows))
            ops.gemm(xs, dz, g, R, 2 * Dc, rows - d, R, 2 * Dc, 2 * Dc, a_mode=1, b_mode=1, a_off=xo + d * R, b_off=d * 2 * Dc,
                     c_off=fg + R * 2 * Dc, accumulate=2, split_k=self._sk(rows))
            # dx_l = dx_{l+1} + dz . W[1]^T (same row) + dz[m] . W[0]^T -> row m - d
            # (dz has max(dilation) zero rows behind its end, so the shifted read covers every row; two launches through
            # a scratch buffer, no accumulate= - that needs an fp32 C - and nothing in place)
            dxt = self._buf("dx_tmp", rows * R, T_)
            ops.gemm(dz, W, dxt, rows, R, 2 * Dc, 2 * Dc, 2 * Dc, R, b_mode=0, a_off=d * 2 * Dc, b_off=fg, addend=dx,
                     addend_off=nxt, ld_add=R)
            ops.gemm(dz, W, dx, rows, R, 2 * Dc, 2 * Dc, 2 * Dc, R, b_mode=0, b_off=fg + R * 2 * Dc, c_off=cur, addend=dxt,
                     ld_add=R)
        if self.scalar_input:        # dW[k] = sum_m s[m - (IFW-1) + k] * dx0[m]
            K = self.IFW
            ops.gemm(self._net_in, dx, g, K, R, rows - (K - 1), 1, R, R, a_mode=1, b_mode=1, b_off=(K - 1) * R,
                     c_off=self._o("causal"), accumulate=2)
        else:
            ops.wavenet_input(self._net_in, None, None, N, T0, R, Q, dx=dx, dw=g, dw_off=self._o("causal"), start=1)
        if dgc is not None:
            dh = self._buf("dgc_h", N * self.gc, f32)
            ops.zero(dh)
            for n, v in enumerate(self._gc_ids):
                ops.colsum(dgc, self.gc, T0, self.gc, dh, x_off=n * T0 * self.gc, out_off=n * self.gc)
                ops.copy3d(dh, g, 1, 1, self.gc, (0, 0), (0, 0), src_off=n * self.gc,
                           dst_off=self._o("gc_embedding") + v * self.gc, accumulate=1)
        return self

    def apply_gradients(self):
        hp = self._hparams
        t = self.global_step + 1
        lr = self.learning_rate_at(self.global_step)
        b1, b2 = hp.adam["beta1"], hp.adam["beta2"]
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        n = self.layout.size
        ops.sumsq(self.flat_g, n, self.scal, out_off=8, work=self._buf("sumsq_work", 1032, torch.float32))
        ops.adam(self.flat_p, self.flat_g, self.flat_m, self.flat_v, n, self.scal[8:], self.gradient_clip, 1.0 / self.world_size,
                 lr_t, b1, b2, 1e-8, shadow=self.flat_s if self.flat_s is not self.flat_p else None)
        self.learning_rate = lr
        self.global_step += 1
        return self

    def read_losses(self):
        self.loss = float(self.scal[0].item())
        return self.loss

    def step(self, audio_inputs=None, global_conditions=None, local_conditions=None):
        if audio_inputs is not None:
            self.initialize(audio_inputs, global_conditions, local_conditions)
        else:                       # the batch of the last call again
            self.initialize_ids(self.ids, *self._last)
        self.backward()
        self.apply_gradients()
        return self.read_losses()

    # ------------------------------------------------------------------ inference
    def predict_proba(self, waveform_ids, global_condition=None):
        """wavenet_simple.py:436-453 / wavenet.py:610-632: float64 softmax of the last position's logits for one
        waveform - mu-law codes, or samples with scalar_input; global_condition: one category id or one vector."""
        w = np.asarray(waveform_ids)
        T0 = int(w.shape[-1])
        if self.scalar_input:
            a = torch.from_numpy(np.ascontiguousarray(w.astype(np.float32).reshape(-1))).to(self.device)
            net_in = self._buf("scalar_in", T0, self.T)
            ops.copy3d(a, net_in, 1, 1, T0, (0, 0), (0, 0))
        else:
            net_in = torch.as_tensor(w.astype(np.int32)).to(self.device).view(1, -1).contiguous()
        gcrows = None
        if self.full:
            if self.lc:
                raise NotImplementedError("predict_proba takes no local condition (wavenet.py:622)")
            gc = None if global_condition is None else np.asarray(global_condition)[None]
            gcrows, _ = self._condition_rows(1, T0, gc, None)
        logits, ow = self._forward(net_in, 1, T0, keep=False, gcrows=gcrows)
        probs = torch.empty(self.Q, dtype=torch.float32, device=self.device)
        ops.wavenet_softmax(logits, self.Q, 1, self.Q, probs, logits_off=(ow - 1) * self.Q)
        return probs

    def generate(self, seed_ids, n_samples, uniforms=None, seed=0, exact=None, fast=True, engine=None, global_conditions=None):
        """Incremental generation (generate_wavenet.py:56-142): seed_ids int [B, n_seed] (or [n_seed]) of mu-law codes,
        n_seed >= receptive field; returns int32 [B, n_seed + n_samples].  uniforms [B, n_samples] in [0,1) drive the
        categorical draws (default: numpy Generator(seed)).  Weights: the fp32 master copy when exact (default in fp32
        mode), else the bf16 shadow (half the bytes streamed per sample)."""
        if self.scalar_input:
            raise NotImplementedError("Incremental generation does not support scalar input yet.")       # wavenet.py:643-645
        if self.lc:
            raise NotImplementedError("Incremental generation takes no local condition (wavenet.py:487)")
        seed_ids = np.atleast_2d(np.asarray(seed_ids, np.int32))
        B, n_seed = seed_ids.shape
        assert n_seed >= self.rf, "seed shorter than the receptive field (%d)" % self.rf
        total = n_seed + int(n_samples)
        if uniforms is None:
            uniforms = np.random.default_rng(seed).random((B, n_samples))
        uniforms = np.atleast_2d(np.asarray(uniforms, np.float32))
        dev = self.device
        ids = torch.zeros(B, total, dtype=torch.int32, device=dev)
        ids[:, :n_seed] = torch.from_numpy(seed_ids).to(dev)
        un = torch.from_numpy(np.ascontiguousarray(uniforms)).to(dev)
        qrows = int(sum(self.dil))
        queues = torch.zeros(B * qrows * self.R, dtype=torch.float32, device=dev)
        dil = torch.tensor(self.dil, dtype=torch.int32, device=dev)
        if exact is None:
            exact = self.T == torch.float32
        W = self.flat_p if exact else self.flat_s
        offs = dict(causal=self._o("causal"), layer0=self._o("fg0"),
                    layer_stride=(self._o("fg1") - self._o("fg0")) if self.L > 1 else 0,
                    dense_in_layer=self._o("dense0") - self._o("fg0"), skip=self._o("skip"), post1=self._o("post1"),
                    post2=self._o("post2"))
        self.last_probs = torch.zeros(B * self.Q, dtype=torch.float32, device=dev)
        extra = self._generator_terms(B, W, global_conditions) if self.full else {}
        if extra:
            fast = False            # conditions / biases: the per-layer kernel
        fgT = deT = None
        ok = self.S % 8 == 0 and self.Q % 8 == 0 and 512 % (self.S // 8) == 0 and 512 % (self.Q // 8) == 0
        if fast and not exact and self.R == self.Dc and self.R in (16, 32) and ok:
            # column-major bf16 shadows of the layer kernels for the single-wave chain (a lane loads its whole column)
            R, Dc = self.R, self.Dc
            fg = torch.stack([self.flat_p[self._o("fg%d" % l):self._o("fg%d" % l) + 2 * R * 2 * Dc].view(2 * R, 2 * Dc).t()
                              for l in range(self.L)])
            de = torch.stack([self.flat_p[self._o("dense%d" % l):self._o("dense%d" % l) + Dc * R].view(Dc, R).t()
                              for l in range(self.L)])
            fgT, deT = fg.contiguous().to(torch.bfloat16), de.contiguous().to(torch.bfloat16)
        if engine is None:          # MFMA chain + concurrent skip waves at the shipped widths, else the single-wave VALU chain
            engine = 2 if (fgT is not None and self.R == 32 and self.S <= 512) else 1
            if engine == 2 and self.S == 512 and self.Q == 256 and B * 5 <= torch.cuda.get_device_properties(dev).multi_processor_count \
                    and self._helper(dev) is not None:
                engine = 3          # + the post-processing products on four helper workgroups per waveform (weights in registers)
        if engine in (2, 3):
            # the MFMA chain keeps the activations in fragment layout between layers: operand slot 8 g + j of a
            # 32-wide K block holds channel 4 g + (j & 3) + 16 (j >> 2) (wavenet.hip, wn_generate_mfma_kernel)
            perm = torch.tensor([4 * (i // 8) + (i % 4) + 16 * ((i % 8) // 4) for i in range(32)], device=dev)
            fgT = torch.cat([fgT[:, :, :32][:, :, perm], fgT[:, :, 32:][:, :, perm]], dim=2).contiguous()
            deT = deT[:, :, perm].contiguous()
        if engine == 3:
            assert self._helper(dev) is not None, "engine 3 needs a stream that runs beside the current one"
            post_x = torch.empty(ops.wavenet_post_floats(B), dtype=torch.float32, device=dev)
            extra = dict(extra, post_x=post_x, helper_stream=self._helper_stream)
        ops.wavenet_generate(W, offs, dil, self.L, self.R, self.Dc, self.S, self.Q, B, n_seed, total, qrows, ids, un, queues,
                             probs=self.last_probs, fgT=fgT, deT=deT, engine=engine, **extra)
        if engine == 3:             # the call's stream takes the helpers' end in: post_x is free behind it
            torch.cuda.current_stream(dev).wait_stream(self._helper_stream)
            self.last_status = extra["post_x"][:1].view(torch.int32)
            if int(self.last_status.item()) != 0:        # (the caller reads the samples next anyway: one wait here)
                raise RuntimeError("ns_wavenet_generate: a hand-over between the chain and its helper workgroups timed out "
                                   "(their workgroups were not resident together?) - the samples are invalid; engine=2 "
                                   "runs without helpers")
        self.last_engine = engine
        self._gen_keep = (fgT, deT, un, queues, dil, extra)  # keep the operands alive until the stream has used them
        return ids

    def _helper(self, dev):
        """The helper kernel's stream: one that was seen to run BESIDE the current stream (ns_streams_concurrent), or None
        (then the helpers would start only after the chain kernel: engine 2 it is)."""
        if getattr(self, "_helper_stream", None) is None:
            s = ops.concurrent_stream(dev)
            self._helper_stream = s if ops.streams_concurrent(torch.cuda.current_stream(dev), s) else False
        return self._helper_stream or None

    def _generator_terms(self, B, W, global_conditions):
        """What the full model's incremental generator adds per layer (wavenet.py:398-437), formed once per call:
        cond [B, L, 2Dc] = h . [gc_filter | gc_gate] + [filter_bias | gate_bias], and the other biases gathered."""
        L, Dc, R, S = self.L, self.Dc, self.R, self.S
        f32 = torch.float32
        out = {}
        if self.gc or self.use_biases:
            cond = self._buf("gen_cond", B * L * 2 * Dc, f32)
            ops.zero(cond)
            if self.gc:
                gcrows, _ = self._condition_rows(B, 1, global_conditions, None)     # [B, gc] in the compute dtype
                h = gcrows
                if h.dtype != W.dtype:
                    h = self._buf("gen_h", B * self.gc, W.dtype)
                    ops.copy3d(gcrows, h, 1, 1, B * self.gc, (0, 0), (0, 0))
                ops.F32_PASSES = self.passes
                for l in range(L):
                    ops.gemm(h, W, cond, B, 2 * Dc, self.gc, self.gc, 2 * Dc, L * 2 * Dc, b_mode=1, b_off=self._o("gc%d" % l),
                             c_off=l * 2 * Dc, **self._bias("fgb%d" % l))
            else:
                assert global_conditions is None, "gc_channels is 0: no global condition expected"
                for l in range(L):
                    ops.copy3d(self.flat_p, cond, B, 1, 2 * Dc, (0, 0), (L * 2 * Dc, 0), src_off=self._o("fgb%d" % l),
                               dst_off=l * 2 * Dc)
            out["cond"] = cond
        else:
            assert global_conditions is None, "gc_channels is 0: no global condition expected"
        if self.use_biases:
            deb = self._buf("gen_deb", L * R, f32)
            for l in range(L):
                ops.copy3d(self.flat_p, deb, 1, 1, R, (0, 0), (0, 0), src_off=self._o("deb%d" % l), dst_off=l * R)
            sb = self._buf("skb_sum", S, f32)
            ops.zero(sb)
            ops.colsum(self.flat_p, S, L, S, sb, x_off=self._o("skb"))
            p1 = self._buf("gen_p1b", S, f32)
            p2 = self._buf("gen_p2b", self.Q, f32)
            ops.copy3d(self.flat_p, p1, 1, 1, S, (0, 0), (0, 0), src_off=self._o("p1b"))
            ops.copy3d(self.flat_p, p2, 1, 1, self.Q, (0, 0), (0, 0), src_off=self._o("p2b"))
            out.update(dense_bias=deb, skip_bias=sb, post1_bias=p1, post2_bias=p2)
        return out


class WaveNetModel(SimpleWaveNet):
    """create_model('wavenet', hparams): neural_speech/models/wavenet.py with its options - use_biases, scalar_input
    (initial_filter_width), global conditioning by category (gc_category_cardinality) or by vector (gc_channels), local
    conditioning (lc_channels).  With every option off (the shipped wavenet.yaml) it is SimpleWaveNet, kernel for kernel.
    0 and None both mean "off" (the shipped yaml writes 0)."""
    full = True

    def _options(self, hp):
        self.use_biases = bool(hp.use_biases)
        self.scalar_input = bool(hp.scalar_input)
        self.IFW = int(hp.initial_filter_width) if self.scalar_input else 2
        self.gc = int(hp.gc_channels or 0)
        self.gc_card = int(hp.gc_category_cardinality or 0) if self.gc else 0
        self.lc = int(hp.lc_channels or 0)
