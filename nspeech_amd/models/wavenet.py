"""simple_wavenet (neural_speech/models/wavenet_simple.py) on the HIP kernels: training graph (initialize /
add_loss / add_optimizer), predict_proba and sample-by-sample generation.

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


def receptive_field(hp):
    """wavenet_simple.py:124-128."""
    fw = hp.filter_width
    return (fw - 1) * sum(dilations(hp)) + 1 + (fw - 1)


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
        assert hp.filter_width == 2 and not hp.use_biases and not hp.scalar_input, "shipped wavenet.yaml only"
        assert not (hp.gc_channels or hp.lc_channels), "conditioning is not built (SURVEY F1)"
        self.device = torch.device(device)
        self.mode = dtype
        self.T = torch.float32 if dtype == "fp32" else torch.bfloat16
        self.passes = 0
        self.world_size = world_size
        self.dil = dilations(hp)
        self.rf = receptive_field(hp)
        self.L = len(self.dil)
        self.Q, self.R, self.Dc, self.S = hp.quantization_channels, hp.residual_channels, hp.dilation_channels, hp.skip_channels
        lay = self.layout = Layout()
        lay.add("causal", (2, self.Q, self.R))
        for i in range(self.L):
            lay.add("fg%d" % i, (2, self.R, 2 * self.Dc))
            lay.add("dense%d" % i, (self.Dc, self.R))
        lay.add("skip", (self.L, self.Dc, self.S))
        lay.add("post1", (self.S, self.S))
        lay.add("post2", (self.S, self.Q))
        n = lay.size
        dev = self.device
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
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

    # ------------------------------------------------------------------ parameters
    def _tf_shapes(self):
        d = OrderedDict()
        d["wavenet/causal_layer/filter"] = (2, self.Q, self.R)
        for i in range(self.L):
            pre = "wavenet/dilated_stack/layer%d/" % i
            d[pre + "filter"] = (2, self.R, self.Dc)
            d[pre + "gate"] = (2, self.R, self.Dc)
            d[pre + "dense"] = (1, self.Dc, self.R)
            d[pre + "skip"] = (1, self.Dc, self.S)
        d["wavenet/postprocessing/postprocess1"] = (1, self.S, self.S)
        d["wavenet/postprocessing/postprocess2"] = (1, self.S, self.Q)
        return d

    def _init_values(self, seed):
        """create_variable = xavier_initializer_conv2d (wavenet_simple.py:12-17): Glorot uniform per variable."""
        rng = np.random.RandomState(seed)
        return OrderedDict((k, glorot_uniform(rng, s)) for k, s in self._tf_shapes().items())

    def _o(self, name):
        return self.layout.off(name)

    def load_numpy_params(self, p):
        host = np.zeros(self.layout.size, np.float32)

        def put(name, arr):
            o = self._o(name)
            host[o:o + arr.size] = np.asarray(arr, np.float32).ravel()
        put("causal", p["wavenet/causal_layer/filter"])
        skips = []
        for i in range(self.L):
            pre = "wavenet/dilated_stack/layer%d/" % i
            put("fg%d" % i, np.concatenate([p[pre + "filter"], p[pre + "gate"]], axis=2))
            put("dense%d" % i, p[pre + "dense"][0])
            skips.append(p[pre + "skip"][0])
        put("skip", np.stack(skips))
        put("post1", p["wavenet/postprocessing/postprocess1"][0])
        put("post2", p["wavenet/postprocessing/postprocess2"][0])
        self.flat_p.copy_(torch.from_numpy(host))
        self.refresh_shadows()

    def _unflatten(self, flat):
        host = flat.detach().float().cpu().numpy()

        def get(name):
            o = self._o(name)
            shp = self.layout.shape(name)
            return host[o:o + int(np.prod(shp))].reshape(shp).copy()
        out = OrderedDict()
        out["wavenet/causal_layer/filter"] = get("causal")
        sk = get("skip")
        for i in range(self.L):
            pre = "wavenet/dilated_stack/layer%d/" % i
            fg = get("fg%d" % i)
            out[pre + "filter"], out[pre + "gate"] = fg[:, :, :self.Dc].copy(), fg[:, :, self.Dc:].copy()
            out[pre + "dense"] = get("dense%d" % i)[None]
            out[pre + "skip"] = sk[i][None]
        out["wavenet/postprocessing/postprocess1"] = get("post1")[None]
        out["wavenet/postprocessing/postprocess2"] = get("post2")[None]
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
            b = torch.zeros(numel, dtype=dtype, device=self.device)
            self._bufs[name] = b
        return b

    # ------------------------------------------------------------------ network
    def _forward(self, ids, N, T0, keep):
        """ids int32 [N, T0] on the device.  Returns (logits fp32 [N*ow, Q], ow).  keep: save what backward needs."""
        T_, W = self.T, self.flat_s
        R, Dc, S, Q, L = self.R, self.Dc, self.S, self.Q, self.L
        rows = N * T0
        ow = T0 - self.rf + 1
        assert ow >= 1, "clip shorter than the receptive field (%d samples)" % self.rf
        ops.F32_PASSES = self.passes
        xs = self._buf("xs", (L + 1) * rows * R if keep else 2 * rows * R, T_)
        outs = self._buf("outs", rows * L * Dc, T_)
        zs = self._buf("zs", (L if keep else 1) * rows * 2 * Dc, torch.float32)
        ops.wavenet_input(ids, self.flat_p, xs, N, T0, R, Q, w_off=self._o("causal"))
        start = 1
        for l, d in enumerate(self.dil):
            xo = (l if keep else l % 2) * rows * R
            xn = ((l + 1) if keep else (l + 1) % 2) * rows * R
            zo = (l if keep else 0) * rows * 2 * Dc
            fg = self._o("fg%d" % l)
            # z[m] = x[m-d] . W[0] + x[m] . W[1]  for rows m >= d
            ops.gemm(xs, W, zs, rows - d, 2 * Dc, R, R, 2 * Dc, 2 * Dc, b_mode=1, a_off=xo, b_off=fg, c_off=zo + d * 2 * Dc)
            ops.gemm(xs, W, zs, rows - d, 2 * Dc, R, R, 2 * Dc, 2 * Dc, b_mode=1, a_off=xo + d * R, b_off=fg + R * 2 * Dc,
                     c_off=zo + d * 2 * Dc, accumulate=1)
            start += d
            ops.wavenet_gate(zs[zo:], rows, Dc, T0, start, out=outs, out_off=l * Dc, ld_out=L * Dc)
            # x_next = out . dense + x
            ops.gemm(outs, W, xs, rows, R, Dc, L * Dc, R, R, b_mode=1, a_off=l * Dc, b_off=self._o("dense%d" % l), c_off=xn,
                     addend=xs, addend_off=xo, ld_add=R)
        # skip sum on the rows the loss uses (t >= rf - 1), relu -> post1 -> relu -> post2
        t1 = self._buf("t1", N * ow * S, T_)
        for n in range(N):
            ops.gemm(outs, W, t1, ow, S, L * Dc, L * Dc, S, S, b_mode=1, a_off=(n * T0 + self.rf - 1) * L * Dc,
                     b_off=self._o("skip"), c_off=n * ow * S, act=ACT_RELU)
        c1 = self._buf("c1", N * ow * S, T_)
        ops.gemm(t1, W, c1, N * ow, S, S, S, S, S, b_mode=1, b_off=self._o("post1"), act=ACT_RELU)
        logits = self._buf("logits", N * ow * Q, torch.float32)
        ops.gemm(c1, W, logits, N * ow, Q, S, S, Q, Q, b_mode=1, b_off=self._o("post2"))
        return logits, ow

    def initialize(self, audio_inputs, global_conditions=None, local_conditions=None):
        """wavenet_simple.py:455-477 + add_loss :479-502: audio float [N, T] in [-1, 1]; runs the forward pass and
        the loss (its gradient wrt the logits comes out of the same kernel)."""
        hp = self._hparams
        audio = np.asarray(audio_inputs, np.float32)
        if audio.ndim == 1:
            audio = audio[None]
        ids = torch.from_numpy(mu_law_encode(audio, self.Q)).to(self.device)
        return self.initialize_ids(ids)

    def initialize_ids(self, ids):
        N, T = ids.shape
        T0 = T - 1
        self.ids = ids.to(self.device, torch.int32).contiguous()
        net_in = self.ids[:, :T0].contiguous()
        self.dims = dict(N=N, T0=T0)
        logits, ow = self._forward(net_in, N, T0, keep=True)
        self.dims["ow"] = ow
        self.targets = self.ids[:, self.rf:].contiguous()                   # [N, ow]
        self.raw_output = logits[:N * ow * self.Q].view(N, ow, self.Q)
        self.scal.zero_()
        self.dlogits = self._buf("dlogits", N * ow * self.Q, self.T)
        ops.wavenet_softmax_ce(logits, self.Q, self.targets, N * ow, self.Q, 1.0 / (N * ow), self.scal, dlogits=self.dlogits,
                               ld_d=self.Q)
        self._net_in = net_in
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
        g.zero_()
        f32 = torch.float32
        t1, c1, outs, xs, zs = B["t1"], B["c1"], B["outs"], B["xs"], B["zs"]
        # post2 / post1
        ops.gemm(c1, self.dlogits, g, S, Q, M, S, Q, Q, a_mode=1, b_mode=1, c_off=self._o("post2"), accumulate=2, split_k=self._sk(M))
        dc1 = self._buf("dc1", M * S, f32)
        ops.gemm(self.dlogits, W, dc1, M, S, Q, Q, Q, S, b_mode=0, b_off=self._o("post2"))
        dp1 = self._buf("dp1", M * S, T_)
        ops.act_bwd(dc1, c1, dp1, M, S, ACT_RELU)
        ops.gemm(t1, dp1, g, S, S, M, S, S, S, a_mode=1, b_mode=1, c_off=self._o("post1"), accumulate=2, split_k=self._sk(M))
        dt1 = self._buf("dt1", M * S, f32)
        ops.gemm(dp1, W, dt1, M, S, S, S, S, S, b_mode=0, b_off=self._o("post1"))
        dsk = self._buf("dsk", M * S, T_)
        ops.act_bwd(dt1, t1, dsk, M, S, ACT_RELU)
        # skip GEMM: d(outs) on the loss rows, and the stacked skip kernels
        douts = self._buf("douts", rows * L * Dc, T_)
        douts.zero_()
        for n in range(N):
            ao = (n * T0 + self.rf - 1) * L * Dc
            ops.gemm(dsk, W, douts, ow, L * Dc, S, S, S, L * Dc, b_mode=0, a_off=n * ow * S, b_off=self._o("skip"), c_off=ao)
            ops.gemm(outs, dsk, g, L * Dc, S, ow, L * Dc, S, S, a_mode=1, b_mode=1, a_off=ao, b_off=n * ow * S,
                     c_off=self._o("skip"), accumulate=2, split_k=self._sk(ow))
        # dilated stack, last layer first; dx of the last residual output is zero (nothing reads it)
        dx = self._buf("dx", 2 * rows * R, T_)
        dx.zero_()
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
            ops.wavenet_gate(zs[zo:], rows, Dc, T0, starts[l + 1], dout=dol, ld_dout=Dc, dz=dz)
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
        ops.wavenet_input(self._net_in, None, None, N, T0, R, Q, dx=dx, dw=g, dw_off=self._o("causal"), start=1)
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

    def step(self, audio_inputs=None):
        if audio_inputs is not None:
            self.initialize(audio_inputs)
        else:
            self.initialize_ids(self.ids)
        self.backward()
        self.apply_gradients()
        return self.read_losses()

    # ------------------------------------------------------------------ inference
    def predict_proba(self, waveform_ids):
        """wavenet_simple.py:436-453: float64 softmax of the last position's logits for one waveform of ids."""
        ids = torch.as_tensor(np.asarray(waveform_ids, np.int32)).to(self.device).view(1, -1).contiguous()
        logits, ow = self._forward(ids, 1, ids.shape[1], keep=False)
        probs = torch.empty(self.Q, dtype=torch.float32, device=self.device)
        ops.wavenet_softmax(logits, self.Q, 1, self.Q, probs, logits_off=(ow - 1) * self.Q)
        return probs

    def generate(self, seed_ids, n_samples, uniforms=None, seed=0, exact=None, fast=True, engine=None):
        """Incremental generation (generate_wavenet.py:56-142): seed_ids int [B, n_seed] (or [n_seed]) of mu-law codes,
        n_seed >= receptive field; returns int32 [B, n_seed + n_samples].  uniforms [B, n_samples] in [0,1) drive the
        categorical draws (default: numpy Generator(seed)).  Weights: the fp32 master copy when exact (default in fp32
        mode), else the bf16 shadow (half the bytes streamed per sample)."""
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
        if engine == 2:
            # the MFMA chain keeps the activations in fragment layout between layers: operand slot 8 g + j of a
            # 32-wide K block holds channel 4 g + (j & 3) + 16 (j >> 2) (wavenet.hip, wn_generate_mfma_kernel)
            perm = torch.tensor([4 * (i // 8) + (i % 4) + 16 * ((i % 8) // 4) for i in range(32)], device=dev)
            fgT = torch.cat([fgT[:, :, :32][:, :, perm], fgT[:, :, 32:][:, :, perm]], dim=2).contiguous()
            deT = deT[:, :, perm].contiguous()
        ops.wavenet_generate(W, offs, dil, self.L, self.R, self.Dc, self.S, self.Q, B, n_seed, total, qrows, ids, un, queues,
                             probs=self.last_probs, fgT=fgT, deT=deT, engine=engine)
        self._gen_keep = (fgT, deT, un, queues, dil)        # keep the operands alive until the stream has used them
        return ids
