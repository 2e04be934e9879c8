"""Tacotron-1 (neural_speech/models/tacotron.py:16-190) on the same HIP kernels as Tacotron-2:
embedding -> prenet -> CBHG(K=16) encoder -> Bahdanau attention with a GRU(256) attention cell
behind the decoder prenet -> Dense(256) -> 2 x residual GRU(256) -> r frames -> post CBHG(K=8)
-> Dense(num_freq); Griffin-Lim audio attribute; Noam learning-rate schedule.

Conv banks (k = 1..16) use the padded layout with 7 / 8 pad rows, so every bank conv is the same
strided GEMM as in Tacotron-2.  Bahdanau attention is the location-sensitive kernel with a zero
location filter.  GRU recurrences run one launch per gate product from this file (skinny GEMMs with
sigmoid / tanh epilogues + small element-wise kernels); this is the reference's CPU "plumbing"
configuration (BASELINE configs[0]), so the time loops are plain Python, not fused kernels.

Forward ops record their backward on a tape; gradients wrt activations are fp32 and accumulate.
"""
import math
import os

import numpy as np
import torch

from .. import ops
from .._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SOFTSIGN, ACT_TANH
from . import params as P_
from .tacotron2 import Tacotron2, _LazyAudio, _round_up


class Act(object):
    """A [N, P, C] activation in the padded layout plus its (lazily created) fp32 gradient."""

    def __init__(self, model, name, N, P, padl, T, C, dtype=None, buf=None):
        self.m, self.name, self.N, self.P, self.padl, self.T, self.C = model, name, N, P, padl, T, C
        self.rows = N * P
        self.buf = buf if buf is not None else model._buf("a:" + name, self.rows * C, dtype or model.T)
        self._grad = None

    @property
    def mask(self):
        return (self.P, self.padl, self.padl + self.T)

    @property
    def grad(self):
        if self._grad is None:
            # zero here already: _buf clears a buffer when it creates it, backward() clears every existing "g:" buffer in
            # one launch before the tape runs
            self._grad = self.m._buf("g:" + self.name, self.rows * self.C, torch.float32)
        return self._grad


class Tacotron(Tacotron2):
    # data parallel: one whole-buffer all-reduce once backward() has been enqueued (parallel.whole_buffer_range)
    _BUCKET_AFTER = {"backward": "all"}

    # Weight gradients of the non-recurrent layers run on a second stream, on operand buffers of their layers' own: ~2.8 ms
    # of k-long split-K products per step leave the main stream's dependent chain and fill what the recurrences (ns_gru_seq /
    # the attention clusters: 4 - 16 CUs, or waiting on their hops) and the small launches leave idle (round 5).
    WG = "taco1"

    # eager: released as the backward pass forms their operands (19.1 ms per step at the benchmark shape); queue: held back
    # for the next recurrence (19.5 - a burst of chip-filling products starves the small launches of the OTHER streams
    # while it lasts, the GRU windows among them, and what is queued behind the last recurrence ends as a serial tail);
    # after: queue, released behind the first window's launch (19.4)
    WG_POLICY = os.environ.get("NS_T1_WG_POLICY", "eager")

    @property
    def queue_groups(self):
        return () if self.WG_POLICY == "eager" else (self.WG,)

    def _defer(self, group, fn, eager=False):
        return Tacotron2._defer(self, group, fn, eager=eager or self.WG_POLICY == "eager")

    def _make_streams(self):
        """The weight-gradient stream and the GRU pipeline's two: three streams that run beside the main one AND beside
        one another (ops.concurrent_streams: four hardware queues, four streams)."""
        if self._side is None:
            self._side, sa, sb = ops.concurrent_streams(self.device, 3)
            self._pipe = (sa, sb)

    def _side_stream(self):
        self._make_streams()
        return Tacotron2._side_stream(self)

    padl, padr = 8, 8     # bank widths up to 16: 'same' needs 7 left / 8 right; the data gradient the mirror
    use_attn_cluster = os.environ.get("NS_TACO1_ATTN_CLUSTER", "1") != "0"     # persistent attention loop (csrc/attn_gru.hip)
    LAYOUT = staticmethod(P_.taco1_layout)
    KW = 1    # Bahdanau = location-sensitive kernel with a 1-tap zero filter

    # num_speakers > 1 (tacotron.py:41-66): a 128-wide projection joins the decoder prenet output as in Tacotron-2
    # (rnn_wrappers.py:28-30); the encoder CBHG takes one per highway layer and one as the BiGRU's initial state
    # (modules.py:157-169, _cbhg below)
    _speaker_width = staticmethod(P_.taco2_speaker_width)

    # ------------------------------------------------------------------ shadows
    def refresh_shadows(self, full=False):
        if full and self.flat_s is not self.flat_p:
            ops.cast2d(self.flat_p, 1, self.layout.size, self.layout.size, self.flat_s, self.layout.size, False)
        if not hasattr(self, "tsh"):
            self.tsh = {}
        hp = self._hparams
        dev, T = self.device, self.T

        def tr(key, name, r0, rows, cols):
            if key not in self.tsh:
                self.tsh[key] = torch.zeros(cols * rows, dtype=T, device=dev)
            ops.cast2d(self.flat_p, rows, cols, cols, self.tsh[key], rows, True, src_off=self._o(name) + r0 * cols)

        def gru(key, scope, cin, H):
            tr(key + "_gT", scope + "/gates/kernel", cin, H, 2 * H)
            tr(key + "_cT", scope + "/candidate/kernel", cin, H, H)

        Dsp = self.Dsp
        for cb in ("encoder_cbhg", "post_cbhg"):
            cin = P_.cbhg_highway_widths(hp.speaker_embed_dim if Dsp and cb == "encoder_cbhg" else 0)[-1]
            for d in ("fw", "bw"):
                gru("%s_%s" % (cb, d), "%s/bidirectional_rnn/%s/gru_cell" % (cb, d), cin, 128)
        A, D, M, E = hp.attention_dim, hp.decoder_dim, hp.num_mels, 256
        gru("gru_1", "decoder/gru_1", D, D)
        gru("gru_2", "decoder/gru_2", D, D)
        # attention GRU: the whole [x | h] kernels transposed (x = prenet output is not hoistable)
        tr("att_gT", "decoder/attention_gru/gates/kernel", 0, 128 + Dsp + A, 2 * A)
        tr("att_cT", "decoder/attention_gru/candidate/kernel", 0, 128 + Dsp + A, A)
        tr("w1cT", "decoder/decoder_prenet/dense_1/kernel", M, E, 256)
        tr("w2T", "decoder/decoder_prenet/dense_2/kernel", 0, 256, 128)
        tr("wqT", "decoder/attention/query_layer/kernel", 0, A, A)
        if "wcl" not in self.tsh:
            self.tsh["wcl"] = torch.zeros(A, dtype=torch.float32, device=dev)      # zero location term
        F = hp.num_freq
        Fp = _round_up(F, 16)
        if "wl_pad" not in self.tsh:
            self.tsh["wl_pad"] = torch.zeros(256 * Fp, dtype=T, device=dev)
            self.tsh["bl_pad"] = torch.zeros(Fp, dtype=torch.float32, device=dev)
        ops.cast2d(self.flat_p, 256, F, F, self.tsh["wl_pad"], Fp, False, src_off=self._o("dense/kernel"))
        ops.cast2d(self.flat_p, 1, F, F, self.tsh["bl_pad"], Fp, False, src_off=self._o("dense/bias"))

    # ------------------------------------------------------------------ schedule (tacotron.py:143-146,186-190)
    def learning_rate_at(self, step):
        hp = self._hparams
        if not hp.decay_learning_rate:
            return hp.initial_learning_rate
        warm = 4000.0
        s = float(step + 1)
        return hp.initial_learning_rate * warm ** 0.5 * min(s * warm ** -1.5, s ** -0.5)

    # ------------------------------------------------------------------ taped layer ops
    def _new(self, name, like, C, dtype=None):
        return Act(self, name, like.N, like.P, like.padl, like.T, C, dtype)

    def _dense(self, name, x, scope, cout, act, mask=True, W=None, woff=None, boff=None, ldw=None):
        """y = act(x . W + b) per row (tf.layers.dense); pad rows forced to 0 when mask."""
        y = self._new(name, x, cout)
        woff = self._o(scope + "/kernel") if woff is None else woff
        boff = self._o(scope + "/bias") if boff is None else boff
        Wb = self._W(self.T) if W is None else W
        rm = (x.P, x.padl, x.padl + x.T, 0) if mask else None
        ops.gemm(x.buf, Wb, y.buf, x.rows, cout, x.C, x.C, cout, cout, b_mode=1, b_off=woff, bias=self.flat_p,
                 bias_off=boff, act=act, row_mask=rm)

        def bwd():
            # (a queued weight gradient reads dpre later: a buffer of the layer's own then)
            dpre = self._buf("t:dpre" + (":" + name if self.overlap_wgrads else ""), x.rows * cout, self.T)
            ops.act_bwd(y.grad, y.buf, dpre, x.rows, cout, act, row_mask=x.mask if mask else None)
            g = self.flat_g

            def wgrad():
                ops.gemm(x.buf, dpre, g, x.C, cout, x.rows, x.C, cout, cout, a_mode=1, b_mode=1, c_off=woff, accumulate=2,
                         split_k=self._splitk(x.rows, x.C, cout))
                ops.colsum(dpre, cout, x.rows, cout, g, out_off=boff)
            self._defer(self.WG, wgrad)
            ops.gemm(dpre, Wb, x.grad, x.rows, x.C, cout, cout, cout, x.C, a_mode=0, b_mode=0, b_off=woff, accumulate=1)
        self._tape.append(bwd)
        return y

    def _conv(self, name, x, scope, k, cout, act, training, into=None):
        """into = (Act, first column): the layer's output IS that column block of the wider activation (a convolution bank's
        concatenation, modules.py:121-128, without a copy either way: BatchNorm writes there, its backward reads the
        block of the wide gradient)."""
        if into is not None:
            wide, col = into
            self._conv_fwd(scope, x.buf, x.C, cout, k, act, x.N, x.T, x.P, "c:" + name, training=training,
                           y_out=(wide.buf, col, wide.C))
            y, dy = None, lambda: (wide.grad, col, wide.C)
        else:
            yb = self._conv_fwd(scope, x.buf, x.C, cout, k, act, x.N, x.T, x.P, "c:" + name, training=training)
            y = Act(self, name, x.N, x.P, x.padl, x.T, cout, buf=yb)
            dy = lambda: y.grad

        def bwd():
            self._conv_bwd(scope, x.buf, dy(), x.C, cout, k, act, x.N, x.T, x.P, "c:" + name, x.grad,
                           dx_accumulate=True, defer=self.WG)
        self._tape.append(bwd)
        return y

    def _concat(self, name, parts):
        C = sum(p.C for p in parts)
        y = self._new(name, parts[0], C)
        off = 0
        for p in parts:
            ops.copy3d(p.buf, y.buf, 1, p.rows, p.C, (0, p.C), (0, C), dst_off=off)
            off += p.C

        def bwd():
            o = 0
            for p in parts:
                ops.copy3d(y.grad, p.grad, 1, p.rows, p.C, (0, C), (0, p.C), src_off=o, accumulate=1)
                o += p.C
        self._tape.append(bwd)
        return y

    def _concat_time_bcast(self, name, x, v):
        """[x | v tiled over time] (modules.py:160-162): v is one row per utterance (an Act with P = 1)."""
        C, Cv = x.C, v.C
        y = self._new(name, x, C + Cv)
        ops.copy3d(x.buf, y.buf, 1, x.rows, C, (0, C), (0, C + Cv))
        ops.copy3d(v.buf, y.buf, x.N, x.P, Cv, (Cv, 0), (x.P * (C + Cv), C + Cv), dst_off=C)

        def bwd():
            ops.copy3d(y.grad, x.grad, 1, x.rows, C, (0, C + Cv), (0, C), accumulate=1)
            for n in range(x.N):        # column sums over the utterance's rows (pad rows carry zero gradient)
                ops.colsum(y.grad, C + Cv, x.P, Cv, v.grad, x_off=n * x.P * (C + Cv) + C, out_off=n * Cv)
        self._tape.append(bwd)
        return y

    def _add(self, name, a, b):
        y = self._new(name, a, a.C)
        ops.copy3d(a.buf, y.buf, 1, a.rows, a.C, (0, a.C), (0, a.C))
        ops.copy3d(b.buf, y.buf, 1, a.rows, a.C, (0, a.C), (0, a.C), accumulate=1)

        def bwd():
            for t in (a, b):
                ops.copy3d(y.grad, t.grad, 1, a.rows, a.C, (0, a.C), (0, a.C), accumulate=1)
        self._tape.append(bwd)
        return y

    def _highway(self, name, x, scope):
        C = x.C
        h = self._new(name + "_h", x, C)
        t = self._new(name + "_t", x, C)
        y = self._new(name, x, C)
        W = self._W(self.T)
        oh, ot = self._o(scope + "/H/kernel"), self._o(scope + "/T/kernel")
        bh, bt = self._o(scope + "/H/bias"), self._o(scope + "/T/bias")
        ops.gemm(x.buf, W, h.buf, x.rows, C, C, C, C, C, b_mode=1, b_off=oh, bias=self.flat_p, bias_off=bh, act=ACT_RELU)
        ops.gemm(x.buf, W, t.buf, x.rows, C, C, C, C, C, b_mode=1, b_off=ot, bias=self.flat_p, bias_off=bt,
                 act=ACT_SIGMOID)
        ops.highway(h.buf, t.buf, x.buf, x.rows * C, y=y.buf)

        def bwd():
            own = (":" + name) if self.overlap_wgrads else ""          # the queued weight gradients read them later
            dh = self._buf("t:hw_dh" + own, x.rows * C, self.T)
            dt_ = self._buf("t:hw_dt" + own, x.rows * C, self.T)
            dx = self._buf("t:hw_dx", x.rows * C, torch.float32)
            ops.highway(h.buf, t.buf, x.buf, x.rows * C, dy=y.grad, dhpre=dh, dtpre=dt_, dx=dx)
            g = self.flat_g

            def wgrad():
                for dpre, ow, ob in ((dh, oh, bh), (dt_, ot, bt)):
                    ops.gemm(x.buf, dpre, g, C, C, x.rows, C, C, C, a_mode=1, b_mode=1, c_off=ow, accumulate=2,
                             split_k=self._splitk(x.rows, C, C))
                    ops.colsum(dpre, C, x.rows, C, g, out_off=ob)
            self._defer(self.WG, wgrad)
            for dpre, ow in ((dh, oh), (dt_, ot)):
                ops.gemm(dpre, W, dx, x.rows, C, C, C, C, C, a_mode=0, b_mode=0, b_off=ow, accumulate=1)
            ops.copy3d(dx, x.grad, 1, x.rows, C, (0, C), (0, C), accumulate=1)
        self._tape.append(bwd)
        return y

    # ---- GRU over time (tf GRUCell); h history lives in out.buf columns [col, col+H)
    use_gru_seq = os.environ.get("NS_GRU_SEQ", "1") != "0"      # persistent whole-sequence kernels where the shape allows

    def _gru_seq(self, tag, x, scope, key, H, lengths, reverse, out, col, h0=None):
        """One GRU direction over the whole sequence (see _gru_group)."""
        self._gru_group(tag, [dict(tag=tag, scope=scope, key=key, reverse=reverse, col=col)], x, H, lengths, out, h0)

    def _gru_group(self, label, dirs, x, H, lengths, out, h0=None):
        """One or two GRU directions (a BiGRU's fw / bw cells, modules.py:172-181) over the same input x, outputs in
        out.buf columns [col, col + H) of each direction.  The input halves of both gate products are hoisted GEMMs; the
        recurrence is ONE persistent launch (ns_gru_seq_*, csrc/gru.hip) where the shape allows - H in {128, 256}, a
        split-bf16 or bf16 mode - and otherwise four launches per step from here.  last_paths["<label>:fwd|bwd"] says which.

        h0: optional Act [N, 1, H], the initial state (modules.py:165-181).  The persistent kernels take it as their
        initial state.  The per-step path's recurrent products read the state history, where the slot in front of a
        sequence's first step holds zeros, so there the initial state enters three ways: h0 . Wg_h is added to the
        input-side gate pre-activations of the first step (t = 0; reversed: t = len(n) - 1), the element-wise kernels
        substitute h0 for h_prev there (h_init), and the weight gradient of the recurrent gate kernel gets the first
        steps' h0^T . dzg on top of the shifted-history product (both paths)."""
        N, P, padl, T = x.N, x.P, x.padl, x.T
        rows, cin, ldh = x.rows, x.C, out.C
        W, g = self._W(self.T), self.flat_g
        hb = out.buf
        h0f = None
        if h0 is not None:
            h0f = self._buf("gru:%s_h0f" % label, N * H, torch.float32)
            ops.copy3d(h0.buf, h0f, 1, N, H, (0, H), (0, H))
        lens = self._host_lengths if lengths is not None else [T] * N
        for dd in dirs:
            tag, scope = dd["tag"], dd["scope"]
            dd["og"], dd["oc"] = self._o(scope + "/gates/kernel"), self._o(scope + "/candidate/kernel")
            dd["bg"], dd["bc"] = self._o(scope + "/gates/bias"), self._o(scope + "/candidate/bias")
            dd["xg"] = self._buf("gru:%s_xg" % tag, rows * 2 * H, torch.float32)
            dd["xc"] = self._buf("gru:%s_xc" % tag, rows * H, torch.float32)
            ops.gemm(x.buf, W, dd["xg"], rows, 2 * H, cin, cin, 2 * H, 2 * H, b_mode=1, b_off=dd["og"], bias=self.flat_p,
                     bias_off=dd["bg"])
            ops.gemm(x.buf, W, dd["xc"], rows, H, cin, cin, H, H, b_mode=1, b_off=dd["oc"], bias=self.flat_p, bias_off=dd["bc"])
            # saved gates: [rows, 2H] / [rows, H] for the step launches; the persistent kernels keep them in an order of their
            # own over whole 16-row groups (ns_gru_seq_params.ru) - the buffers serve either
            srows = max(rows, (N + 15) // 16 * 16 * T)
            dd["ru"] = self._buf("gru:%s_ru" % tag, srows * 2 * H, torch.float32)
            dd["cc"] = self._buf("gru:%s_c" % tag, srows * H, torch.float32)
            dd["rh"] = self._buf("gru:%s_rh" % tag, rows * H, self.T)
            dd["gT"], dd["cT"] = self.tsh[dd["key"] + "_gT"], self.tsh[dd["key"] + "_cT"]
            dd["first"] = [int(lens[n]) - 1 if dd["reverse"] else 0 for n in range(N)]

        def params(dd, bwd):
            kw = {}
            if bwd:
                kw = dict(dh=(out.grad, dd["col"]), ld_dh=ldh, dzg=dd["dzg"], dzc=dd["dzc"])
                if h0 is not None:
                    kw.update(dh_init=dd["dh0"], ld_dhi=H)
            # every [N*P, .] array is addressed as row n * P + padl + t: the arrays start at row 0
            return ops.gru_seq_params(hb, N, T, H, P, padl, dd["reverse"], lengths, dd["xg"], dd["xc"], dd["gT"], dd["cT"],
                                      (W, dd["og"] + cin * 2 * H), 2 * H, (W, dd["oc"] + cin * H), H, (hb, dd["col"]), ldh,
                                      dd["ru"], dd["cc"], dd["rh"], h_init=h0f, ld_hi=H, **kw)

        def persistent(bwd):
            if not self.use_gru_seq or len(dirs) > 2:
                return None
            pp = [params(dd, bwd) for dd in dirs]
            p1 = pp[1] if len(pp) > 1 else None
            return (pp[0], p1) if ops.gru_seq_supported(pp[0], p1, backward=bwd) else None

        pp = persistent(False)
        if pp is not None:
            work = self._buf("gru:%s_work" % label, ops.gru_seq_work_floats(pp[0]), torch.float32)
            ops.gru_seq("fwd", pp[0], pp[1], work)
            self._status_words[(label, "fwd")] = work
            self.last_paths["%s:fwd" % label] = "seq"
        else:
            self.last_paths["%s:fwd" % label] = "step"
            for dd in dirs:
                self._gru_steps_fwd(dd, N, P, padl, T, H, ldh, lengths, hb, h0, h0f)

        def bwd():
            self._flush_deferred()          # queued weight gradients run beside this recurrence
            for dd in dirs:
                dd["dzg"] = self._buf("gru:%s_dzg" % dd["tag"], rows * 2 * H, self.T)
                dd["dzc"] = self._buf("gru:%s_dzc" % dd["tag"], rows * H, self.T)
                ops.zero(dd["dzg"])          # the hoisted products below run over every row: pad rows must hold zeros
                ops.zero(dd["dzc"])
                if h0 is not None:
                    dd["dh0"] = self._buf("gru:%s_dh0" % dd["tag"], N * H, torch.float32)
            pb = persistent(True)
            if pb is not None:
                work = self._buf("gru:%s_work" % label, ops.gru_seq_work_floats(pb[0]), torch.float32)
                ops.gru_seq("bwd", pb[0], pb[1], work)
                self._status_words[(label, "bwd")] = work
                self.last_paths["%s:bwd" % label] = "seq"
            else:
                if self.last_paths.get("%s:fwd" % label) == "seq":
                    raise RuntimeError("GRU %s: the persistent forward kernel's saved gates are private to ns_gru_seq_bwd" % label)
                self.last_paths["%s:bwd" % label] = "step"
                for dd in dirs:
                    self._gru_steps_bwd(dd, N, P, padl, T, H, ldh, lengths, hb, out.grad, cin, h0f)
            for dd in reversed(dirs):
                self._gru_hoisted_bwd(dd, x, H, hb, ldh, h0)
        self._tape.append(bwd)

    # ---- the decoder's two residual GRUs as a pipeline in time (round 5)
    use_gru_pipeline = os.environ.get("NS_GRU_PIPE", "1") != "0"
    GRU_PIPE_CHUNKS = int(os.environ.get("NS_GRU_PIPE_CHUNKS", "4"))

    def _gru_pair(self, x1, H):
        """y1 = x1 + GRU_1(x1); y2 = y1 + GRU_2(y1) (tacotron.py:72-76) under teacher forcing.  Step s of GRU_2 needs only
        step s of GRU_1, and a persistent GRU(256) launch occupies 8 of the 256 CUs - so the time axis is cut into
        GRU_PIPE_CHUNKS windows and GRU_2 runs window c on a second stream while GRU_1 runs window c + 1 on the first
        (each window is one ns_gru_seq launch on the rows [t0, t1) with the state at t0 - 1 as its initial state; GRU_2's
        input products of a window are a batched GEMM over the window's rows).  Backward the same in reverse: GRU_2's
        window c, its input gradients, then GRU_1's window c on the other stream while GRU_2 runs window c - 1; the
        gradient wrt a window's initial state is added to the history gradient of the row in front of it.  The hoisted
        weight gradients stay whole-sequence products.  Returns None when the persistent kernels do not take the shape
        (the caller then runs the two GRUs one after the other)."""
        N, P, padl, T = x1.N, x1.P, x1.padl, x1.T
        rows, D = x1.rows, x1.C
        if not self.use_gru_seq or D != H or T < 2 * self.GRU_PIPE_CHUNKS or self.device.type != "cuda":
            return None
        W, T_ = self._W(self.T), self.T
        f32 = torch.float32
        nch = self.GRU_PIPE_CHUNKS
        edges = [T * c // nch for c in range(nch + 1)]
        wins = [(edges[c], edges[c + 1]) for c in range(nch)]
        N16 = (N + 15) // 16 * 16
        h1 = self._new("dec_h1", x1, H); y1 = self._new("dec_y1", x1, H)
        h2 = self._new("dec_h2", x1, H); y2 = self._new("dec_y2", x1, H)
        gr = []
        for k, (tag, xin) in enumerate((("gru_1", x1), ("gru_2", y1))):
            scope = "decoder/" + tag
            dd = dict(tag=tag, key=tag, reverse=False, col=0, x=xin, out=(h1, h2)[k])
            dd["og"], dd["oc"] = self._o(scope + "/gates/kernel"), self._o(scope + "/candidate/kernel")
            dd["bg"], dd["bc"] = self._o(scope + "/gates/bias"), self._o(scope + "/candidate/bias")
            dd["xg"] = self._buf("gru:%s_xg" % tag, rows * 2 * H, f32)
            dd["xc"] = self._buf("gru:%s_xc" % tag, rows * H, f32)
            dd["ru"] = self._buf("gru:%s_ru" % tag, max(rows, N16 * T) * 2 * H, f32)
            dd["cc"] = self._buf("gru:%s_c" % tag, max(rows, N16 * T) * H, f32)
            dd["rh"] = self._buf("gru:%s_rh" % tag, rows * H, T_)
            dd["gT"], dd["cT"] = self.tsh[tag + "_gT"], self.tsh[tag + "_cT"]
            dd["hi"] = [None] + [self._buf("gru:%s_hi%d" % (tag, c), N * H, f32) for c in range(1, nch)]
            dd["work"] = self._buf("gru:%s_work" % tag, 1, f32)
            gr.append(dd)

        def params(dd, c, bwd):
            t0, t1 = wins[c]
            hb = dd["out"].buf
            kw = {}
            if bwd:
                kw = dict(dh=(dd["out"].grad, 0), ld_dh=H, dzg=dd["dzg"], dzc=dd["dzc"])
                if c > 0:
                    kw.update(dh_init=dd["dhi"], ld_dhi=H)
            # a window = the same arrays with the first row moved to t0: every [N*P, .] array is addressed as row n * P + padl + t
            return ops.gru_seq_params(hb, N, t1 - t0, H, P, padl + t0, False, None, dd["xg"], dd["xc"], dd["gT"], dd["cT"],
                                      (W, dd["og"] + D * 2 * H), 2 * H, (W, dd["oc"] + D * H), H, (hb, 0), H,
                                      (dd["ru"], N16 * t0 * 2 * H), (dd["cc"], N16 * t0 * H), dd["rh"],
                                      h_init=dd["hi"][c], ld_hi=H, **kw)

        for dd in gr:
            p0 = params(dd, 1, False)
            if not ops.gru_seq_supported(p0, None, backward=False):
                return None
            dd["work"] = self._buf("gru:%s_work" % dd["tag"], ops.gru_seq_work_floats(p0), f32)
        g1, g2 = gr
        ops.zero_many((h1.buf, h2.buf, y1.buf))       # (y1's pad rows: the windows below write the steps' rows only)
        ops.gemm(x1.buf, W, g1["xg"], rows, 2 * H, D, D, 2 * H, 2 * H, b_mode=1, b_off=g1["og"], bias=self.flat_p, bias_off=g1["bg"])
        ops.gemm(x1.buf, W, g1["xc"], rows, H, D, D, H, H, b_mode=1, b_off=g1["oc"], bias=self.flat_p, bias_off=g1["bc"])
        # (a batched product takes no bias: GRU_2's input products add onto rows that hold the biases)
        ops.copy3d(self.flat_p, g2["xg"], 1, rows, 2 * H, (0, 0), (0, 2 * H), src_off=g2["bg"])
        ops.copy3d(self.flat_p, g2["xc"], 1, rows, H, (0, 0), (0, H), src_off=g2["bc"])
        # Both chains run on streams of their own that were probed to run side by side (_make_streams): HIP deals its
        # hardware queues to streams in turn, and a single side stream shares the main stream's queue whenever the
        # process has made a multiple of the queue count before it (measured: the seventh stream of a process - 22.0
        # instead of 20.8 ms per step, the two chains serialised and the events on top).  The main stream only waits.
        self._make_streams()
        main = torch.cuda.current_stream(self.device)
        sa, sb = self._pipe

        def hand_over(src, dst):            # dst goes on behind everything src holds so far
            ev = torch.cuda.Event()
            ev.record(src)
            dst.wait_event(ev)

        def window(dd, c, direction):
            t0, _ = wins[c]
            if direction == "fwd" and c > 0:        # the state in front of the window is its initial state
                ops.copy3d(dd["out"].buf, dd["hi"][c], N, 1, H, (P * H, 0), (H, 0), src_off=(padl + t0 - 1) * H)
            ops.gru_seq(direction, params(dd, c, direction == "bwd"), None, dd["work"])
            if direction == "bwd" and c > 0:        # ... and its gradient belongs to the row in front of the window
                ops.copy3d(dd["dhi"], dd["out"].grad, N, 1, H, (H, 0), (P * H, 0), dst_off=(padl + t0 - 1) * H, accumulate=1)

        def rows_of(c, C):                  # (I, J, strides, offset) of a window's rows in a [N*P, C] array
            t0, t1 = wins[c]
            return N, t1 - t0, (P * C, C), (padl + t0) * C

        hand_over(main, sa)
        for c in range(nch):
            with torch.cuda.stream(sa):
                window(g1, c, "fwd")
            hand_over(sa, sb)
            with torch.cuda.stream(sb):
                I, J, st, off = rows_of(c, H)
                ops.copy3d(x1.buf, y1.buf, I, J, H, st, st, src_off=off, dst_off=off)
                ops.copy3d(h1.buf, y1.buf, I, J, H, st, st, src_off=off, dst_off=off, accumulate=1)
                ops.gemm(y1.buf, W, g2["xg"], J, 2 * H, H, H, 2 * H, 2 * H, b_mode=1, a_off=off, b_off=g2["og"], c_off=2 * off,
                         accumulate=1, batch=N, batch_strides=(P * H, 0, P * 2 * H))
                ops.gemm(y1.buf, W, g2["xc"], J, H, H, H, H, H, b_mode=1, a_off=off, b_off=g2["oc"], c_off=off,
                         accumulate=1, batch=N, batch_strides=(P * H, 0, P * H))
                window(g2, c, "fwd")
        hand_over(sb, main)
        ops.copy3d(y1.buf, y2.buf, 1, rows, H, (0, H), (0, H))
        ops.copy3d(h2.buf, y2.buf, 1, rows, H, (0, H), (0, H), accumulate=1)
        for dd in gr:
            self._status_words[(dd["tag"], "fwd")] = dd["work"]
            self.last_paths["%s:fwd" % dd["tag"]] = "seq"
        self.last_paths["gru_pair"] = "pipelined x%d" % nch

        def bwd():
            if self.WG_POLICY != "after":
                self._flush_deferred()      # the weight gradients queued so far (the post CBHG's) run beside the windows below
            main = torch.cuda.current_stream(self.device)
            # y2 = y1 + h2
            ops.copy3d(y2.grad, y1.grad, 1, rows, H, (0, H), (0, H), accumulate=1)
            ops.copy3d(y2.grad, h2.grad, 1, rows, H, (0, H), (0, H), accumulate=1)
            for dd in gr:
                dd["dzg"] = self._buf("gru:%s_dzg" % dd["tag"], rows * 2 * H, T_)
                dd["dzc"] = self._buf("gru:%s_dzc" % dd["tag"], rows * H, T_)
                dd["dhi"] = self._buf("gru:%s_dhi" % dd["tag"], N * H, f32)
            ops.zero_many((g1["dzg"], g1["dzc"], g2["dzg"], g2["dzc"]))      # pad rows must hold zeros for the hoisted products
            hand_over(main, sa)
            for c in range(nch - 1, -1, -1):
                I, J, st, off = rows_of(c, H)
                with torch.cuda.stream(sa):
                    window(g2, c, "bwd")
                    # the window's gradient wrt GRU_2's input y1
                    ops.gemm(g2["dzg"], W, y1.grad, J, H, 2 * H, 2 * H, 2 * H, H, a_mode=0, b_mode=0, a_off=2 * off,
                             b_off=g2["og"], c_off=off, accumulate=1, batch=N, batch_strides=(P * 2 * H, 0, P * H))
                    ops.gemm(g2["dzc"], W, y1.grad, J, H, H, H, H, H, a_mode=0, b_mode=0, a_off=off, b_off=g2["oc"], c_off=off,
                             accumulate=1, batch=N, batch_strides=(P * H, 0, P * H))
                hand_over(sa, sb)
                with torch.cuda.stream(sb):
                    ops.copy3d(y1.grad, h1.grad, I, J, H, st, st, src_off=off, dst_off=off, accumulate=1)      # y1 = x1 + h1
                    window(g1, c, "bwd")
                if c == nch - 1 and self.WG_POLICY == "after":
                    self._flush_deferred()  # behind the first windows' launches
            hand_over(sb, main)
            ops.copy3d(y1.grad, x1.grad, 1, rows, H, (0, H), (0, H), accumulate=1)
            for dd in (g2, g1):
                self._status_words[(dd["tag"], "bwd")] = dd["work"]
                self.last_paths["%s:bwd" % dd["tag"]] = "seq"
                self._gru_hoisted_bwd(dd, dd["x"], H, dd["out"].buf, H, None, input_grads=dd is g1)
        self._tape.append(bwd)
        return y2

    def _gru_hoisted_bwd(self, dd, x, H, hb, ldh, h0=None, input_grads=True):
        """What the backward pass of one GRU direction hoists out of its time loop, given the gate gradients dzg / dzc of
        every step: weight gradients - x parts, h parts (h_prev = history shifted by one row), biases - and the gradient
        wrt the input (input_grads=False: the caller has formed it already)."""
        N, P, padl = x.N, x.P, x.padl
        rows, cin = x.rows, x.C
        W, g = self._W(self.T), self.flat_g
        sk = self._splitk
        og, oc, dzg, dzc, rh = dd["og"], dd["oc"], dd["dzg"], dd["dzc"], dd["rh"]

        def wgrad():        # queued (self.WG): every operand is a buffer of this GRU's own and stays as it is until the join
            ops.gemm(x.buf, dzg, g, cin, 2 * H, rows, cin, 2 * H, 2 * H, a_mode=1, b_mode=1, c_off=og, accumulate=2,
                     split_k=sk(rows, cin, 2 * H))
            ops.gemm(x.buf, dzc, g, cin, H, rows, cin, H, H, a_mode=1, b_mode=1, c_off=oc, accumulate=2,
                     split_k=sk(rows, cin, H))
            if dd["reverse"]:
                ops.gemm(hb, dzg, g, H, 2 * H, rows - 1, ldh, 2 * H, 2 * H, a_mode=1, b_mode=1, a_off=ldh + dd["col"],
                         c_off=og + cin * 2 * H, accumulate=2, split_k=sk(rows, H, 2 * H))
            else:
                ops.gemm(hb, dzg, g, H, 2 * H, rows - 1, ldh, 2 * H, 2 * H, a_mode=1, b_mode=1, a_off=dd["col"], b_off=2 * H,
                         c_off=og + cin * 2 * H, accumulate=2, split_k=sk(rows, H, 2 * H))
            ops.gemm(rh, dzc, g, H, H, rows, H, H, H, a_mode=1, b_mode=1, c_off=oc + cin * H, accumulate=2,
                     split_k=sk(rows, H, H))
            if h0 is not None:          # the first steps' h_prev was h0
                dz0 = self._buf("gru:%s_dz0" % dd["tag"], N * 2 * H, self.T)
                for n in range(N):
                    ops.copy3d(dzg, dz0, 1, 1, 2 * H, (0, 0), (0, 0), src_off=(n * P + padl + dd["first"][n]) * 2 * H,
                               dst_off=n * 2 * H)
                ops.gemm(h0.buf, dz0, g, H, 2 * H, N, H, 2 * H, 2 * H, a_mode=1, b_mode=1, c_off=og + cin * 2 * H,
                         accumulate=2)
            ops.colsum(dzg, 2 * H, rows, 2 * H, g, out_off=dd["bg"])
            ops.colsum(dzc, H, rows, H, g, out_off=dd["bc"])
        self._defer(self.WG, wgrad)
        if h0 is not None:
            # what is left in the carry is the gradient wrt the initial state
            ops.copy3d(dd["dh0"], h0.grad, 1, N, H, (0, H), (0, H), accumulate=1)
        if input_grads:
            ops.gemm(dzg, W, x.grad, rows, cin, 2 * H, 2 * H, 2 * H, cin, a_mode=0, b_mode=0, b_off=og, accumulate=1)
            ops.gemm(dzc, W, x.grad, rows, cin, H, H, H, cin, a_mode=0, b_mode=0, b_off=oc, accumulate=1)

    def _gru_steps_fwd(self, dd, N, P, padl, T, H, ldh, lengths, hb, h0, h0f):
        """The launch-per-step form of one direction: two gate products + two element-wise kernels per step."""
        xg, xc, ru, cc, rh, gT, cT, col, reverse = (dd[k] for k in ("xg", "xc", "ru", "cc", "rh", "gT", "cT", "col", "reverse"))
        hi = {}
        if h0 is not None:
            hi = dict(h_init=(h0f, 0), hi_sn=H, reverse=reverse, T=T)
            sg = self._buf("gru:%s_h0g" % dd["tag"], N * 2 * H, torch.float32)
            ops.gemm(h0.buf, gT, sg, N, 2 * H, H, H, H, 2 * H)
            for n in range(N):
                ops.copy3d(sg, xg, 1, 1, 2 * H, (0, 0), (0, 0), src_off=n * 2 * H,
                           dst_off=(n * P + padl + dd["first"][n]) * 2 * H, accumulate=1)
        dd["hi"] = hi
        for t in (range(T - 1, -1, -1) if reverse else range(T)):
            row, prow = padl + t, padl + (t + 1 if reverse else t - 1)
            ops.gemm(hb, gT, ru, N, 2 * H, H, P * ldh, H, P * 2 * H, a_off=prow * ldh + col, c_off=row * 2 * H,
                     act=ACT_SIGMOID, addend=xg, addend_off=row * 2 * H, ld_add=P * 2 * H)
            ops.gru_pointwise(0, hb, N, H, t, lengths, ru=(ru, row * 2 * H), ru_sn=P * 2 * H,
                              h_prev=(hb, prow * ldh + col), hp_sn=P * ldh, out=(rh, row * H), out_sn=P * H, **hi)
            ops.gemm(rh, cT, cc, N, H, H, P * H, H, P * H, a_off=row * H, c_off=row * H, act=ACT_TANH, addend=xc,
                     addend_off=row * H, ld_add=P * H)
            ops.gru_pointwise(1, hb, N, H, t, lengths, ru=(ru, row * 2 * H), ru_sn=P * 2 * H, c=(cc, row * H),
                              c_sn=P * H, h_prev=(hb, prow * ldh + col), hp_sn=P * ldh, out=(hb, row * ldh + col),
                              out_sn=P * ldh, **hi)

    def _gru_steps_bwd(self, dd, N, P, padl, T, H, ldh, lengths, hb, dh, cin, h0f):
        ru, cc, col, reverse, dzg, dzc, og, oc = (dd[k] for k in ("ru", "cc", "col", "reverse", "dzg", "dzc", "og", "oc"))
        W = self._W(self.T)
        hi = dd.get("hi")
        if hi is None:
            hi = dict(h_init=(h0f, 0), hi_sn=H, reverse=reverse, T=T) if h0f is not None else {}
        carry = self._buf("gru:carry", N * H, torch.float32)
        drh = self._buf("gru:drh", N * H, torch.float32)
        ops.zero(carry)
        for t in (range(T) if reverse else range(T - 1, -1, -1)):
            row, prow = padl + t, padl + (t + 1 if reverse else t - 1)
            # dh = the recurrent part (carry) + the gradient wrt this step's output where the step is valid; past the
            # length the carry passes unchanged
            ops.gru_pointwise(2, hb, N, H, t, lengths, ru=(ru, row * 2 * H), ru_sn=P * 2 * H, c=(cc, row * H),
                              c_sn=P * H, h_prev=(hb, prow * ldh + col), hp_sn=P * ldh, out=(dzc, row * H),
                              out_sn=P * H, dzg=(dzg, row * 2 * H), dzg_sn=P * 2 * H, dh=(carry, 0), dh_sn=H,
                              carry=(carry, 0), carry_sn=H, dh_add=(dh, row * ldh + col), dha_sn=P * ldh, **hi)
            ops.gemm(dzc, W, drh, N, H, H, P * H, H, H, a_off=row * H, b_off=oc + cin * H)
            ops.gru_pointwise(3, hb, N, H, t, lengths, ru=(ru, row * 2 * H), ru_sn=P * 2 * H,
                              h_prev=(hb, prow * ldh + col), hp_sn=P * ldh, dzg=(dzg, row * 2 * H),
                              dzg_sn=P * 2 * H, dh=(drh, 0), dh_sn=H, carry=(carry, 0), carry_sn=H, **hi)
            ops.gemm(dzg, W, carry, N, H, 2 * H, P * 2 * H, 2 * H, H, a_off=row * 2 * H, b_off=og + cin * 2 * H,
                     accumulate=1)
        if "dh0" in dd:
            ops.copy3d(carry, dd["dh0"], 1, N, H, (0, H), (0, H))

    def _cbhg(self, name, x, lengths, scope, K, proj, training, spk=None):
        """spk: the speaker embedding rows (Act [N, 1, speaker_embed_dim]) for the encoder CBHG, modules.py:157-169."""
        bank = self._new(name + "_bank", x, 128 * K)        # the K bank outputs side by side: every layer writes its own block
        for k in range(1, K + 1):
            self._conv("%s_b%d" % (name, k), x, "%s/conv_bank/conv1d_%d" % (scope, k), k, 128, ACT_RELU, training,
                       into=(bank, (k - 1) * 128))
        y = bank
        for i, size in enumerate(proj[:-1]):      # each reads the bank, the last one wins
            y = self._conv("%s_p%d" % (name, i + 1), bank, "%s/proj_%d" % (scope, i + 1), 3, size, ACT_RELU, training)
        y = self._conv("%s_p%d" % (name, len(proj)), y, "%s/proj_%d" % (scope, len(proj)), 3, proj[-1], ACT_NONE,
                       training)
        hw = self._add(name + "_res", y, x)
        if hw.C != 128:
            hw = self._dense(name + "_dense", hw, scope + "/dense", 128, ACT_NONE)
        for i in range(4):
            if spk is not None:
                sp = self._dense("%s_sp%d" % (name, i), spk, "%s/highway_%d/dense" % (scope, i), hw.C, ACT_SOFTSIGN, mask=False)
                hw = self._concat_time_bcast("%s_hs%d" % (name, i), hw, sp)
            hw = self._highway("%s_hw%d" % (name, i), hw, "%s/highway_%d/highway" % (scope, i))
        h0 = self._dense(name + "_h0", spk, scope + "/dense", 128, ACT_SOFTSIGN, mask=False) if spk is not None else None
        out = self._new(name + "_out", hw, 256)
        ops.zero(out.buf)
        dirs = [dict(tag="%s_%s" % (name, d), scope="%s/bidirectional_rnn/%s/gru_cell" % (scope, d), key="%s_%s" % (scope, d),
                     reverse=d == "bw", col=di * 128) for di, d in enumerate(("fw", "bw"))]
        self._gru_group(name + "_gru", dirs, hw, 128, lengths, out, h0)
        return out

    def _speaker_rows(self, N):
        """tacotron.py:41-48: speaker_embed[speaker_ids] as an Act [N, 1, speaker_embed_dim] (None for one speaker)."""
        if not self.Dsp:
            return None
        sd = self._hparams.speaker_embed_dim
        e = Act(self, "spk_e", N, 1, 0, 1, sd)
        ops.embedding_fwd(self.speaker_ids, self.flat_p, e.buf, N, 1, 1, 0, sd, self.n_speakers,
                          table_off=self._o("speaker/speaker_embed"))
        self._tape.append(lambda: ops.embedding_bwd(self.speaker_ids, e.grad, self.flat_g, N, 1, 1, 0, sd, self.n_speakers,
                                                    dtable_off=self._o("speaker/speaker_embed")))
        return e

    def initialize(self, text_inputs, input_lengths, speaker_ids=None, mel_targets=None, linear_targets=None):
        if linear_targets is not None:
            return Tacotron2.initialize(self, text_inputs, input_lengths, speaker_ids, mel_targets, linear_targets)
        self._set_inputs(text_inputs, input_lengths, speaker_ids)
        self.is_training = False
        self.mel_targets = self.linear_targets = None
        return self.forward_infer()

    # ------------------------------------------------------------------ free-running synthesis
    def _gru_step_infer(self, x, x_off, x_sn, K, gT, cT, bg, bc, xc, H, hprev_off, out, out_off, out_sn, out2, out2_off,
                        out2_sn, ru, cc, N):
        """One GRUCell step on an explicit [input | h_prev] row (x) with whole-kernel transposes."""
        xin = K - H
        ops.gemm(x, gT, ru, N, 2 * H, K, x_sn, K, 2 * H, a_off=x_off, bias=self.flat_p, bias_off=bg, act=ACT_SIGMOID)
        ops.copy3d(x, xc, N, 1, xin, (x_sn, 0), (K, 0), src_off=x_off)
        ops.gru_pointwise(0, x, N, H, 0, None, ru=(ru, 0), ru_sn=2 * H, h_prev=(x, x_off + xin), hp_sn=x_sn,
                          out=(xc, xin), out_sn=K)
        ops.gemm(xc, cT, cc, N, H, K, K, K, H, bias=self.flat_p, bias_off=bc, act=ACT_TANH)
        ops.gru_pointwise(1, x, N, H, 0, None, ru=(ru, 0), ru_sn=2 * H, c=(cc, 0), c_sn=H, h_prev=(x, x_off + xin),
                          hp_sn=x_sn, out=(out, out_off), out_sn=out_sn, out2=(out2, out2_off) if out2 is not None else None,
                          out2_sn=out2_sn)

    def forward_infer(self):
        """tacotron.py with linear_targets=None: TacoTestHelper feedback for max_iters steps (Q7),
        BatchNorm on moving statistics; sets mel_outputs, linear_outputs, alignments and audio."""
        hp = self._hparams
        T_ = self.T
        N, Ti = self.inputs.shape
        r, M, F = hp.outputs_per_step, hp.num_mels, hp.num_freq
        S = int(hp.max_iters)
        To = S * r
        Fp = _round_up(F, 16)
        A, D, E = hp.attention_dim, hp.decoder_dim, 256
        Pi, Po, S1 = Ti + self.padl + self.padr, To + self.padl + self.padr, S + 2
        sig = ("infer1", N, Ti, S)
        if sig != self._sig:
            self._bufs.clear()
            self._sig = sig
        self._tape = []
        ops.F32_PASSES = self.passes_fwd
        W, o, tsh, buf = self._W(T_), self._o, self.tsh, self._buf
        # whole-kernel transposes for the [input | h] steps of the two decoder GRUs
        for key, scope in (("gru_1", "decoder/gru_1"), ("gru_2", "decoder/gru_2")):
            for suf, nm, cols in (("_gTf", "/gates/kernel", 2 * D), ("_cTf", "/candidate/kernel", D)):
                if key + suf not in tsh:
                    tsh[key + suf] = torch.zeros(cols * 2 * D, dtype=T_, device=self.device)
                ops.cast2d(self.flat_p, 2 * D, cols, cols, tsh[key + suf], 2 * D, True, src_off=o(scope + nm))
        for key, nm, rows, cols in (("w1T_full", "decoder/decoder_prenet/dense_1/kernel", M + E, 256),
                                    ("wprojT", "decoder/attention_projection/kernel", A + E, D),
                                    ("woutT", "decoder/output_projection/kernel", D, M * r)):
            if key not in tsh:
                tsh[key] = torch.zeros(rows * cols, dtype=T_, device=self.device)
            ops.cast2d(self.flat_p, rows, cols, cols, tsh[key], rows, True, src_off=o(nm))
        lengths = self.input_lengths
        emb = Act(self, "emb", N, Pi, self.padl, Ti, hp.embedding_dim)
        ops.embedding_fwd(self.inputs, self.flat_p, emb.buf, N, Ti, Pi, self.padl, hp.embedding_dim, self.vocab,
                          table_off=o("embedding/embedding"))
        pn = list(hp.encoder_prenet)
        x = self._dense("pre1", emb, "prenet/dense_1", pn[0], ACT_RELU)
        x = self._dense("pre2", x, "prenet/dense_2", pn[1], ACT_RELU)
        spk = self._speaker_rows(N)
        enc = self._cbhg("enc", x, lengths, "encoder_cbhg", hp.encoder_cbhg_banks, list(hp.encoder_cbhg_bank_sizes), False,
                         spk=spk)
        Dsp = self.Dsp
        XI = 128 + Dsp
        spk_dec = self._dense("spk_dec", spk, "decoder/dense", Dsp, ACT_SOFTSIGN, mask=False) if Dsp else None
        keys = buf("keys", N * Pi * A, torch.float32)
        ops.gemm(enc.buf, W, keys, N * Pi, A, E, E, A, A, b_mode=1, b_off=o("attention_decoder/memory_layer/kernel"))
        Tia = _round_up(Ti, 8)
        keys_t = buf("keys_t", N * A * Tia, torch.float32)
        ops.keys_transpose(keys, keys_t, N, Ti, Tia, Pi, self.padl, A)
        XP, XA, HC, X1 = M + E, XI + A, A + E, 2 * D
        xp = buf("i_xp", N * S1 * XP, T_); p1 = buf("i_p1", N * 256, T_)
        xa = buf("i_xa", N * S1 * XA, T_); xc = buf("i_xc", N * max(XA, X1), T_)
        hc = buf("i_hc", N * S1 * HC, T_)
        g1 = buf("i_g1", N * S1 * X1, T_); g2 = buf("i_g2", N * S1 * X1, T_)     # [input | h_prev] rows of the GRUs
        y1 = buf("i_y1", N * D, T_); y2 = buf("i_y2", N * D, T_); hh = buf("i_hh", N * D, T_)
        ru = buf("i_ru", N * 2 * max(A, D), torch.float32); cc = buf("i_cc", N * max(A, D), torch.float32)
        q = buf("i_q", N * A, torch.float32); al = buf("i_al", N * S1 * Tia, torch.float32)
        er = buf("i_er", N * Tia, torch.float32); dec = buf("i_dec", N * S1 * M * r, torch.float32)
        for b in (xp, xa, hc, g1, g2, al):
            ops.zero(b)
        if Dsp:
            ops.copy3d(spk_dec.buf, xa, N, S1, Dsp, (Dsp, 0), (S1 * XA, XA), dst_off=128)
        ov = o("decoder/attention/attention_v")
        for s in range(S):
            sl, nx = s + 1, s + 2
            ops.gemm(xp, tsh["w1T_full"], p1, N, 256, XP, S1 * XP, XP, 256, a_off=sl * XP, bias=self.flat_p,
                     bias_off=o("decoder/decoder_prenet/dense_1/bias"), act=ACT_RELU)
            ops.gemm(p1, tsh["w2T"], xa, N, 128, 256, 256, 256, S1 * XA, c_off=sl * XA, bias=self.flat_p,
                     bias_off=o("decoder/decoder_prenet/dense_2/bias"), act=ACT_RELU)
            self._gru_step_infer(xa, sl * XA, S1 * XA, XA, tsh["att_gT"], tsh["att_cT"], o("decoder/attention_gru/gates/bias"),
                                 o("decoder/attention_gru/candidate/bias"), xc, A, 0, hc, sl * HC, S1 * HC, xa,
                                 nx * XA + XI, S1 * XA, ru, cc, N)
            ops.gemm(hc, tsh["wqT"], q, N, A, A, S1 * HC, A, A, a_off=sl * HC)
            ops.attention_step(hc, N, Ti, Pi, self.padl, Tia, A, E, self.KW, lengths, keys_t, enc.buf, (q, 0), A,
                               (al, s * Tia), (al, sl * Tia), S1 * Tia, (hc, sl * HC + A), S1 * HC, (xp, nx * XP + M),
                               S1 * XP, tsh["wcl"], (self.flat_p, ov), er)
            # x1 = Dense([h_att | ctx]) -> residual GRU 1 -> residual GRU 2 -> r frames
            ops.gemm(hc, tsh["wprojT"], g1, N, D, HC, S1 * HC, HC, S1 * X1, a_off=sl * HC, c_off=sl * X1, bias=self.flat_p,
                     bias_off=o("decoder/attention_projection/bias"))
            self._gru_step_infer(g1, sl * X1, S1 * X1, X1, tsh["gru_1_gTf"], tsh["gru_1_cTf"], o("decoder/gru_1/gates/bias"),
                                 o("decoder/gru_1/candidate/bias"), xc, D, 0, hh, 0, D, g1, nx * X1 + D, S1 * X1, ru, cc, N)
            ops.copy3d(g1, y1, N, 1, D, (S1 * X1, 0), (D, 0), src_off=sl * X1)
            ops.copy3d(hh, y1, N, 1, D, (D, 0), (D, 0), accumulate=1)
            ops.copy3d(y1, g2, N, 1, D, (D, 0), (S1 * X1, 0), dst_off=sl * X1)
            self._gru_step_infer(g2, sl * X1, S1 * X1, X1, tsh["gru_2_gTf"], tsh["gru_2_cTf"], o("decoder/gru_2/gates/bias"),
                                 o("decoder/gru_2/candidate/bias"), xc, D, 0, hh, 0, D, g2, nx * X1 + D, S1 * X1, ru, cc, N)
            ops.copy3d(y1, y2, N, 1, D, (D, 0), (D, 0))
            ops.copy3d(hh, y2, N, 1, D, (D, 0), (D, 0), accumulate=1)
            ops.gemm(y2, tsh["woutT"], dec, N, M * r, D, D, D, S1 * M * r, c_off=sl * M * r, bias=self.flat_p,
                     bias_off=o("decoder/output_projection/bias"))
            ops.copy3d(dec, xp, N, 1, M, (S1 * M * r, 0), (S1 * XP, 0), src_off=sl * M * r + (r - 1) * M, dst_off=nx * XP)
        melp = Act(self, "mel_pad", N, Po, self.padl, To, M)
        ops.copy3d(dec, melp.buf, N, To, M, (S1 * M * r, M), (Po * M, M), src_off=M * r, dst_off=self.padl * M)
        post = self._cbhg("post", melp, None, "post_cbhg", hp.post_cbhg_banks, list(hp.post_cbhg_bank_sizes) + [M], False)
        lin = buf("lin_out", N * Po * Fp, torch.float32)
        ops.gemm(post.buf, tsh["wl_pad"], lin, N * Po, Fp, 256, 256, Fp, Fp, b_mode=1, bias=tsh["bl_pad"])
        self._tape = []
        self.dims = dict(N=N, Ti=Ti, To=To, S=S, Pi=Pi, Po=Po, Fp=Fp)
        self.mel_outputs = dec[:N * S1 * M * r].view(N, S1, M * r)[:, 1:S + 1].reshape(N, To, M)
        self.decoder_outputs = self.mel_outputs
        self.linear_outputs = lin[:N * Po * Fp].view(N, Po, Fp)[:, self.padl:self.padl + To, :F]
        self.alignments = al[:N * S1 * Tia].view(N, S1, Tia)[:, 1:S + 1, :Ti].permute(0, 2, 1)
        self.audio = _LazyAudio(self)           # tacotron.py:107, vocoded when read
        return self

    # ------------------------------------------------------------------ forward (training)
    def forward_train(self):
        hp = self._hparams
        T_ = self.T
        N, Ti = self.inputs.shape
        To = self.mel_targets.shape[1]
        r, M, F = hp.outputs_per_step, hp.num_mels, hp.num_freq
        assert To % r == 0
        S = To // r
        Fp = _round_up(F, 16)
        A, D, E = hp.attention_dim, hp.decoder_dim, 256
        Pi, Po, S1 = Ti + self.padl + self.padr, To + self.padl + self.padr, S + 1
        sig = ("train1", N, Ti, To)
        if sig != self._sig:
            self._bufs.clear()
            self._sig = sig
        self.dims = dict(N=N, Ti=Ti, To=To, S=S, Pi=Pi, Po=Po, Fp=Fp)
        self._tape = []
        ops.F32_PASSES = self.passes_fwd
        W, g = self._W(T_), self.flat_g
        o = self._o
        lengths = self.input_lengths

        # ---- encoder: embedding -> prenet -> CBHG (tacotron.py:38-62)
        emb = Act(self, "emb", N, Pi, self.padl, Ti, hp.embedding_dim)
        ops.embedding_fwd(self.inputs, self.flat_p, emb.buf, N, Ti, Pi, self.padl, hp.embedding_dim, self.vocab,
                          table_off=o("embedding/embedding"))
        self._tape.append(lambda: ops.embedding_bwd(self.inputs, emb.grad, g, N, Ti, Pi, self.padl, hp.embedding_dim,
                                                    self.vocab, dtable_off=o("embedding/embedding")))
        pn = list(hp.encoder_prenet)
        x = self._dense("pre1", emb, "prenet/dense_1", pn[0], ACT_RELU)
        x = self._dense("pre2", x, "prenet/dense_2", pn[1], ACT_RELU)
        spk = self._speaker_rows(N)
        enc = self._cbhg("enc", x, lengths, "encoder_cbhg", hp.encoder_cbhg_banks, list(hp.encoder_cbhg_bank_sizes), True,
                         spk=spk)
        self._enc = enc
        Dsp = self.Dsp
        self._spk_dec = self._dense("spk_dec", spk, "decoder/dense", Dsp, ACT_SOFTSIGN, mask=False) if Dsp else None

        # ---- attention memory
        keys = self._buf("keys", N * Pi * A, torch.float32)
        om = o("attention_decoder/memory_layer/kernel")
        ops.gemm(enc.buf, W, keys, N * Pi, A, E, E, A, A, b_mode=1, b_off=om)
        Tia = _round_up(Ti, 8)
        keys_t = self._buf("keys_t", N * A * Tia, torch.float32)
        ops.keys_transpose(keys, keys_t, N, Ti, Tia, Pi, self.padl, A)

        # ---- attention RNN over all steps (teacher forced): prenet (-> | speaker projection) -> GRU(A) -> Bahdanau
        XI = 128 + Dsp                            # the GRU's input width
        XA, HC = XI + A, A + E
        fr = self._buf("dec_fr", N * S1 * M, T_)
        if S > 1:
            ops.copy3d(self.mel_targets, fr, N, S - 1, M, (To * M, r * M), (S1 * M, M), src_off=(r - 1) * M, dst_off=2 * M)
        f1 = self._buf("dec_f1", N * S1 * 256, torch.float32)
        w1, w2 = o("decoder/decoder_prenet/dense_1/kernel"), o("decoder/decoder_prenet/dense_2/kernel")
        b1, b2 = o("decoder/decoder_prenet/dense_1/bias"), o("decoder/decoder_prenet/dense_2/bias")
        ops.gemm(fr, W, f1, N * S1, 256, M, M, 256, 256, b_mode=1, b_off=w1, bias=self.flat_p, bias_off=b1)
        p1 = self._buf("dec_p1", N * S1 * 256, T_)
        xa = self._buf("dec_xa", N * S1 * XA, T_)     # [p2 | speaker projection | h_prev]
        xc = self._buf("dec_xc", N * S1 * XA, T_)     # [p2 | speaker projection | r*h_prev]
        hc = self._buf("dec_hc", N * S1 * HC, T_)     # [h | ctx]
        ru = self._buf("dec_ru", N * S1 * 2 * A, torch.float32)
        cc = self._buf("dec_cc", N * S1 * A, torch.float32)
        q = self._buf("dec_q", N * S1 * A, torch.float32)
        al = self._buf("dec_al", N * S1 * Tia, torch.float32)
        al_t = self._buf("dec_al_t", N * S1 * Tia, T_)
        er = self._buf("dec_eraw", N * Tia, torch.float32)
        ops.zero_many((xa, xc, hc, al))
        if Dsp:                                   # rnn_wrappers.py:28-30: the same projection in every step's input
            ops.copy3d(self._spk_dec.buf, xa, N, S1, Dsp, (Dsp, 0), (S1 * XA, XA), dst_off=128)
        tsh = self.tsh
        ag, ac_ = o("decoder/attention_gru/gates/kernel"), o("decoder/attention_gru/candidate/kernel")
        abg, abc = o("decoder/attention_gru/gates/bias"), o("decoder/attention_gru/candidate/bias")
        ov = o("decoder/attention/attention_v")
        # The whole loop as ONE persistent launch (csrc/attn_gru.hip) where the shape allows: the shipped widths,
        # T_in <= 256, 8 N workgroups resident.  Projected-memory form: pv = values . W1c, so the loop yields the
        # next step's prenet layer directly and the 256-wide contexts are formed after it by one product per utterance.
        self._attn_args = None
        if self.use_attn_cluster:
            wg_, wc_ = (W, ag), (W, ac_)
            pvb = self._buf("dec_pv", N * Pi * 256, T_)
            args = dict(dtype=ops.dt(hc), N=N, S=S, Ti=Ti, Pi=Pi, padl_i=self.padl, Tia=Tia, A=A, E=E, D1=256, D2=128,
                        lengths=lengths, keys=keys, pv=pvb, f1=f1, w2=(W, w2), wg=wg_, wc=wc_,
                        wq=(W, o("decoder/attention/query_layer/kernel")), b2=(self.flat_p, b2), bg=(self.flat_p, abg),
                        bc=(self.flat_p, abc), v=(self.flat_p, ov), p1=p1, xa=xa, xc=xc, hc=hc, ru=ru, cc=cc, q=q, align=al,
                        align_t=al_t, Dsp=Dsp)
            if ops.taco1_attn_cluster_supported(**args):
                self._attn_args = args
        if self._attn_args is not None:
            if Dsp:      # the candidate kernel's operand rows [p2 | speaker projection | r * h]: the kernel fills the outer parts
                ops.copy3d(self._spk_dec.buf, xc, N, S1, Dsp, (Dsp, 0), (S1 * XA, XA), dst_off=128)
            ops.gemm(enc.buf, W, pvb, N * Pi, 256, E, E, 256, 256, b_mode=1, b_off=w1 + M * 256)
            cw = self._buf("attn_cluster_work", ops.taco1_attn_cluster_work_floats(**self._attn_args), torch.float32)
            ops.taco1_attn_cluster("fwd", cw, **self._attn_args)
            self._status_words[("attn", "fwd")] = cw
            # contexts of all steps: hc[n, :, A:] = align[n] . values[n] (slot 0: a zero alignment row)
            ops.gemm(al_t, enc.buf, hc, S1, E, Ti, Tia, E, HC, b_mode=1, b_off=self.padl * E, c_off=A, batch=N,
                     batch_strides=(S1 * Tia, Pi * E, S1 * HC))
            self.last_paths["attn:fwd"] = "cluster"
        else:
            self.last_paths["attn:fwd"] = "step"
            for s in range(S):
                sl, pv = s + 1, s
                ops.gemm(hc, tsh["w1cT"], p1, N, 256, E, S1 * HC, E, S1 * 256, a_off=pv * HC + A, c_off=sl * 256, act=ACT_RELU,
                         addend=f1, addend_off=sl * 256, ld_add=S1 * 256)
                ops.gemm(p1, tsh["w2T"], xa, N, 128, 256, S1 * 256, 256, S1 * XA, a_off=sl * 256, c_off=sl * XA,
                         bias=self.flat_p, bias_off=b2, act=ACT_RELU)
                ops.copy3d(xa, xc, N, 1, XI, (S1 * XA, 0), (S1 * XA, 0), src_off=sl * XA, dst_off=sl * XA)
                ops.gemm(xa, tsh["att_gT"], ru, N, 2 * A, XA, S1 * XA, XA, S1 * 2 * A, a_off=sl * XA, c_off=sl * 2 * A,
                         bias=self.flat_p, bias_off=abg, act=ACT_SIGMOID)
                ops.gru_pointwise(0, xa, N, A, s, None, ru=(ru, sl * 2 * A), ru_sn=S1 * 2 * A, h_prev=(xa, sl * XA + XI),
                                  hp_sn=S1 * XA, out=(xc, sl * XA + XI), out_sn=S1 * XA)
                ops.gemm(xc, tsh["att_cT"], cc, N, A, XA, S1 * XA, XA, S1 * A, a_off=sl * XA, c_off=sl * A, bias=self.flat_p,
                         bias_off=abc, act=ACT_TANH)
                ops.gru_pointwise(1, xa, N, A, s, None, ru=(ru, sl * 2 * A), ru_sn=S1 * 2 * A, c=(cc, sl * A), c_sn=S1 * A,
                                  h_prev=(xa, sl * XA + XI), hp_sn=S1 * XA, out=(hc, sl * HC), out_sn=S1 * HC,
                                  out2=(xa, (sl + 1) * XA + XI) if s + 1 < S else None, out2_sn=S1 * XA)
                ops.gemm(hc, tsh["wqT"], q, N, A, A, S1 * HC, A, S1 * A, a_off=sl * HC, c_off=sl * A)
                ops.attention_step(hc, N, Ti, Pi, self.padl, Tia, A, E, self.KW, lengths, keys_t, enc.buf, (q, sl * A), S1 * A,
                                   (al, pv * Tia), (al, sl * Tia), S1 * Tia, (hc, sl * HC + A), S1 * HC, None, 0,
                                   tsh["wcl"], (self.flat_p, ov), er)
            ops.copy3d(al, al_t, 1, N * S1, Tia, (0, Tia), (0, Tia))
        hcA = Act(self, "hc", N, S1, 1, S, HC, buf=hc)
        self._tape.append(self._attention_backward)

        # ---- projection, residual GRUs, output projection (tacotron.py:69-76)
        x1 = self._dense("attproj", hcA, "decoder/attention_projection", D, ACT_NONE, mask=True)
        y2 = self._gru_pair(x1, D) if self.use_gru_pipeline else None
        if y2 is None:
            h1 = self._new("dec_h1", x1, D); ops.zero(h1.buf)
            self._gru_seq("gru_1", x1, "decoder/gru_1", "gru_1", D, None, False, h1, 0)
            y1 = self._add("dec_y1", x1, h1)
            h2 = self._new("dec_h2", y1, D); ops.zero(h2.buf)
            self._gru_seq("gru_2", y1, "decoder/gru_2", "gru_2", D, None, False, h2, 0)
            y2 = self._add("dec_y2", y1, h2)
        decA = Act(self, "dec_out", N, S1, 1, S, M * r, dtype=torch.float32)
        op_, ob_ = o("decoder/output_projection/kernel"), o("decoder/output_projection/bias")
        ops.gemm(y2.buf, W, decA.buf, y2.rows, M * r, D, D, M * r, M * r, b_mode=1, b_off=op_, bias=self.flat_p, bias_off=ob_)
        dec = decA.buf

        def out_proj_bwd():
            dd = self._buf("t:ddec", y2.rows * M * r, T_)
            ops.copy3d(decA.grad, dd, 1, y2.rows, M * r, (0, M * r), (0, M * r))     # slot-0 rows are zero
            ops.gemm(y2.buf, dd, g, D, M * r, y2.rows, D, M * r, M * r, a_mode=1, b_mode=1, c_off=op_, accumulate=2,
                     split_k=self._splitk(y2.rows, D, M * r))
            ops.colsum(dd, M * r, y2.rows, M * r, g, out_off=ob_)
            ops.gemm(dd, W, y2.grad, y2.rows, D, M * r, M * r, M * r, D, a_mode=0, b_mode=0, b_off=op_, accumulate=1)
        self._tape.append(out_proj_bwd)

        # ---- post CBHG + linear head (tacotron.py:92-98)
        melp = Act(self, "mel_pad", N, Po, self.padl, To, M)
        ops.copy3d(dec, melp.buf, N, To, M, (S1 * M * r, M), (Po * M, M), src_off=M * r, dst_off=self.padl * M)
        self._tape.append(lambda: ops.copy3d(melp.grad, decA.grad, N, To, M, (Po * M, M), (S1 * M * r, M),
                                             src_off=self.padl * M, dst_off=M * r, accumulate=1))
        post = self._cbhg("post", melp, None, "post_cbhg", hp.post_cbhg_banks, list(hp.post_cbhg_bank_sizes) + [M], True)
        lin = self._buf("lin_out", N * Po * Fp, torch.float32)
        ops.gemm(post.buf, tsh["wl_pad"], lin, N * Po, Fp, 256, 256, Fp, Fp, b_mode=1, bias=tsh["bl_pad"])

        self._state = dict(keys=keys, keys_t=keys_t, fr=fr, f1=f1, p1=p1, xa=xa, xc=xc, hc=hc, ru=ru, cc=cc, q=q, al=al,
                           al_t=al_t, hcA=hcA, y2=y2, decA=decA, melp=melp, post=post, lin=lin, Tia=Tia)
        self.mel_outputs = dec[:N * S1 * M * r].view(N, S1, M * r)[:, 1:].reshape(N, To, M)
        self.decoder_outputs = self.mel_outputs
        self.linear_outputs = lin[:N * Po * Fp].view(N, Po, Fp)[:, self.padl:self.padl + To, :F]
        self.alignments = al[:N * S1 * Tia].view(N, S1, Tia)[:, 1:, :Ti].permute(0, 2, 1)
        self.audio = _LazyAudio(self)           # tacotron.py:107, vocoded when read
        return self

    # ------------------------------------------------------------------ loss + backward
    def backward(self):
        hp = self._hparams
        T_ = self.T
        ops.DETERMINISTIC_SPLITK = self.deterministic
        d, st = self.dims, self._state
        N, Ti, To, S, Pi, Po, Fp = d["N"], d["Ti"], d["To"], d["S"], d["Pi"], d["Po"], d["Fp"]
        r, M, F = hp.outputs_per_step, hp.num_mels, hp.num_freq
        A, D, E = hp.attention_dim, hp.decoder_dim, 256
        S1, Tia = S + 1, st["Tia"]
        XA, HC = 128 + A, A + E
        g = self.flat_g
        # activation gradients accumulate: start from zero (one launch per 24 buffers)
        ops.zero_many([g, self.scal] + [b for k, b in self._bufs.items() if k.startswith("g:")])
        ops.F32_PASSES = self.passes_bwd
        W, o, tsh = self._W(T_), self._o, self.tsh
        sk = self._splitk
        self._x16_cache = {}                # bf16 copies of layer inputs made in this pass, by source (Tacotron2._conv_bwd)

        # ---- losses (tacotron.py:124-133): mel on the decoder output, linear with the 3 kHz band
        n_prio = int(3000 / (hp.sample_rate * 0.5) * F)
        self._n_prio = n_prio
        decA, melp, post, lin = st["decA"], st["melp"], st["post"], st["lin"]
        dlin = self._buf("d_lin", N * Po * Fp, T_)
        ops.l1_loss(lin, Fp, self.linear_targets, dlin, Fp, N, To, Po, self.padl, F, n_prio, 0.5 / (N * To * F),
                    0.5 / (N * To * n_prio), self.scal, acc_off=2)
        # mel loss gradient lands directly in the decoder-output gradient ([N, S1*r, M] row view)
        ops.l1_loss(decA.buf, M, self.mel_targets, decA.grad, M, N, To, S1 * r, r, M, 0, 1.0 / (N * To * M), 0.0,
                    self.scal, acc_off=0)
        # linear head
        dwl = self._buf("d_wl_pad", 256 * Fp, torch.float32)
        ops.zero(dwl)
        rows_o = N * Po
        ops.gemm(post.buf, dlin, dwl, 256, Fp, rows_o, 256, Fp, Fp, a_mode=1, b_mode=1, accumulate=2,
                 split_k=sk(rows_o, 256, Fp))
        ops.copy3d(dwl, g, 1, 256, F, (0, Fp), (0, F), dst_off=o("dense/kernel"), accumulate=1)
        ops.colsum(dlin, Fp, rows_o, F, g, out_off=o("dense/bias"))
        ops.gemm(dlin, tsh["wl_pad"], post.grad, rows_o, 256, Fp, Fp, Fp, 256, a_mode=0, b_mode=0, accumulate=1)
        # ---- everything else: the tape, newest first
        while self._tape:
            self._tape.pop()()
        self._join_deferred()               # the queued weight gradients are in flat_g behind this
        self._tick("backward")

    def _attention_backward(self):
        """Backward through the attention RNN loop (prenet -> GRU -> Bahdanau), newest step first."""
        hp = self._hparams
        T_ = self.T
        d, st = self.dims, self._state
        N, Ti, S, Pi = d["N"], d["Ti"], d["S"], d["Pi"]
        M = hp.num_mels
        A, E = hp.attention_dim, 256
        S1, Tia = S + 1, st["Tia"]
        Dsp = self.Dsp
        XI = 128 + Dsp
        XA, HC = XI + A, A + E
        g, W, o, tsh, sk = self.flat_g, self._W(T_), self._o, self.tsh, self._splitk
        enc, lengths = self._enc, self.input_lengths
        keys, keys_t, fr, p1, xa, xc, hc, ru, cc, q, al, al_t = (st[k] for k in (
            "keys", "keys_t", "fr", "p1", "xa", "xc", "hc", "ru", "cc", "q", "al", "al_t"))
        dhc = st["hcA"].grad
        rows = N * S1
        buf = self._buf
        dzg = buf("att_dzg", rows * 2 * A, T_); dzc = buf("att_dzc", rows * A, T_)
        dp2 = buf("att_dp2", rows * 128, T_)
        df1 = buf("att_df1", (rows + 1) * 256, T_)          # (+ one zero row read by the shifted context-gradient product)
        dq = buf("att_dq", rows * A, T_); de = buf("att_de", rows * Tia, torch.float32)
        dctx_t = buf("att_dctx_t", rows * E, T_)
        carry_h = buf("att_carry_h", N * A, torch.float32)
        dctx_carry = buf("att_dctx_carry", N * E, torch.float32)
        drh = buf("att_drh", N * A, torch.float32)
        tmp128 = buf("att_tmp128", N * 128, torch.float32)
        gk = buf("att_gk", N * Tia * 8, torch.float32)
        ops.zero_many((dzg, dzc, dp2, df1, dq, de, dctx_t, carry_h, gk))
        da = buf("att_da", N * Tia, torch.float32)
        ag, ac_ = o("decoder/attention_gru/gates/kernel"), o("decoder/attention_gru/candidate/kernel")
        w1, w2 = o("decoder/decoder_prenet/dense_1/kernel"), o("decoder/decoder_prenet/dense_2/kernel")
        wq, ov = o("decoder/attention/query_layer/kernel"), o("decoder/attention/attention_v")
        if self._attn_args is not None:
            # persistent form: da0 = dhc[:, :, A:] . values^T for all steps at once (the part of d(align) that needs no
            # recurrence), the launch, then the context gradients of all steps: dctx[s] = dhc[s, A:] + df1[s+1] . W1c^T
            da0 = buf("att_da0", rows * Tia, torch.float32)
            if T_ == torch.float32:
                src, lda, a0 = dhc, HC, A
            else:
                src, lda, a0 = buf("att_dhc_ctx", rows * E, T_), E, 0
                ops.copy3d(dhc, src, 1, rows, E, (0, HC), (0, E), src_off=A)
            ops.gemm(src, enc.buf, da0, S1, Ti, E, lda, E, Tia, b_mode=0, a_off=a0, b_off=self.padl * E, batch=N,
                     batch_strides=(S1 * lda, Pi * E, S1 * Tia))
            args = dict(self._attn_args)
            args.update(dhc=dhc, da0=da0, df1=df1, dp2=dp2, dzg=dzg, dzc=dzc, dq=dq, de=de)
            cw = buf("attn_cluster_work_b", ops.taco1_attn_cluster_work_floats(**args), torch.float32)
            ops.taco1_attn_cluster("bwd", cw, **args)
            self._status_words[("attn", "bwd")] = cw
            dctx32 = buf("att_dctx32", rows * E, torch.float32)
            ops.copy3d(dhc, dctx32, 1, rows, E, (0, HC), (0, E), src_off=A)
            ops.gemm(df1, W, dctx32, rows, E, 256, 256, 256, E, a_mode=0, b_mode=0, a_off=256, b_off=w1 + M * 256, accumulate=1)
            if T_ == torch.float32:
                dctx_t = dctx32
            else:
                ops.copy3d(dctx32, dctx_t, 1, rows, E, (0, E), (0, E))
            self.last_paths["attn:bwd"] = "cluster"
        else:
            self.last_paths["attn:bwd"] = "step"
            for s in range(S - 1, -1, -1):
                sl, pv = s + 1, s
                last = s == S - 1
                ops.attention_step_bwd(hc, N, Ti, Pi, self.padl, Tia, A, E, self.KW, lengths, keys, keys_t, enc.buf,
                                       (q, sl * A), S1 * A, (al, sl * Tia), (al, pv * Tia), S1 * Tia, (dhc, sl * HC + A), S1 * HC,
                                       None if last else dctx_carry, gk, da, 0 if last else 1, (dq, sl * A), S1 * A,
                                       (de, sl * Tia), (dctx_t, sl * E), S1 * E, tsh["wcl"], (self.flat_p, ov))
                # dh_total = dhc[:, :A] + carry + dq . Wq^T
                ops.copy3d(dhc, carry_h, N, 1, A, (S1 * HC, 0), (A, 0), src_off=sl * HC, accumulate=1)
                ops.gemm(dq, W, carry_h, N, A, A, S1 * A, A, A, a_off=sl * A, b_off=wq, accumulate=1)
                ops.gru_pointwise(2, xa, N, A, s, None, ru=(ru, sl * 2 * A), ru_sn=S1 * 2 * A, c=(cc, sl * A), c_sn=S1 * A,
                                  h_prev=(xa, sl * XA + XI), hp_sn=S1 * XA, out=(dzc, sl * A), out_sn=S1 * A,
                                  dzg=(dzg, sl * 2 * A), dzg_sn=S1 * 2 * A, dh=(carry_h, 0), dh_sn=A, carry=(carry_h, 0), carry_sn=A)
                ops.gemm(dzc, W, drh, N, A, A, S1 * A, A, A, a_off=sl * A, b_off=ac_ + XI * A)
                ops.gru_pointwise(3, xa, N, A, s, None, ru=(ru, sl * 2 * A), ru_sn=S1 * 2 * A, h_prev=(xa, sl * XA + XI),
                                  hp_sn=S1 * XA, dzg=(dzg, sl * 2 * A), dzg_sn=S1 * 2 * A, dh=(drh, 0), dh_sn=A,
                                  carry=(carry_h, 0), carry_sn=A)
                ops.gemm(dzg, W, carry_h, N, A, 2 * A, S1 * 2 * A, 2 * A, A, a_off=sl * 2 * A, b_off=ag + XI * 2 * A, accumulate=1)
                # dp2pre = (dzg . Wg[:128]^T + dzc . Wc[:128]^T) * (p2 > 0)
                ops.gemm(dzg, W, tmp128, N, 128, 2 * A, S1 * 2 * A, 2 * A, 128, a_off=sl * 2 * A, b_off=ag,
                         gate=xa, gate_off=sl * XA, ld_gate=S1 * XA)
                ops.gemm(dzc, W, tmp128, N, 128, A, S1 * A, A, 128, a_off=sl * A, b_off=ac_, accumulate=1,
                         gate=xa, gate_off=sl * XA, ld_gate=S1 * XA)
                ops.copy3d(tmp128, dp2, N, 1, 128, (128, 0), (S1 * 128, 0), dst_off=sl * 128)
                ops.gemm(dp2, W, df1, N, 256, 128, S1 * 128, 128, S1 * 256, a_off=sl * 128, b_off=w2, c_off=sl * 256,
                         gate=p1, gate_off=sl * 256, ld_gate=S1 * 256)
                if s > 0:
                    ops.gemm(df1, W, dctx_carry, N, E, 256, S1 * 256, 256, E, a_off=sl * 256, b_off=w1 + M * 256)
        if Dsp:
            # the speaker projection sits in rows 128 .. 128 + Dsp of both GRU kernels in every slot: its gradient is the
            # sum over an utterance's slots of dzg . Wg[128:XI]^T + dzc . Wc[128:XI]^T
            dsp_rows = buf("att_dspk", rows * Dsp, torch.float32)
            ops.gemm(dzg, W, dsp_rows, rows, Dsp, 2 * A, 2 * A, 2 * A, Dsp, a_mode=0, b_mode=0, b_off=ag + 128 * 2 * A)
            ops.gemm(dzc, W, dsp_rows, rows, Dsp, A, A, A, Dsp, a_mode=0, b_mode=0, b_off=ac_ + 128 * A, accumulate=1)
            for n in range(N):
                ops.colsum(dsp_rows, Dsp, S1, Dsp, self._spk_dec.grad, x_off=n * S1 * Dsp, out_off=n * Dsp)
        # ---- hoisted weight gradients
        ops.gemm(fr, df1, g, M, 256, rows, M, 256, 256, a_mode=1, b_mode=1, c_off=w1, accumulate=2, split_k=sk(rows, M, 256))
        ops.gemm(hc, df1, g, E, 256, rows - 1, HC, 256, 256, a_mode=1, b_mode=1, a_off=A, b_off=256, c_off=w1 + M * 256,
                 accumulate=2, split_k=sk(rows, E, 256))
        ops.colsum(df1, 256, rows, 256, g, out_off=o("decoder/decoder_prenet/dense_1/bias"))
        ops.gemm(p1, dp2, g, 256, 128, rows, 256, 128, 128, a_mode=1, b_mode=1, c_off=w2, accumulate=2, split_k=sk(rows, 256, 128))
        ops.colsum(dp2, 128, rows, 128, g, out_off=o("decoder/decoder_prenet/dense_2/bias"))
        ops.gemm(xa, dzg, g, XA, 2 * A, rows, XA, 2 * A, 2 * A, a_mode=1, b_mode=1, c_off=ag, accumulate=2, split_k=sk(rows, XA, 2 * A))
        ops.colsum(dzg, 2 * A, rows, 2 * A, g, out_off=o("decoder/attention_gru/gates/bias"))
        ops.gemm(xc, dzc, g, XA, A, rows, XA, A, A, a_mode=1, b_mode=1, c_off=ac_, accumulate=2, split_k=sk(rows, XA, A))
        ops.colsum(dzc, A, rows, A, g, out_off=o("decoder/attention_gru/candidate/bias"))
        ops.gemm(hc, dq, g, A, A, rows, HC, A, A, a_mode=1, b_mode=1, c_off=wq, accumulate=2, split_k=sk(rows, A, A))
        # attention: sums over all steps
        dkeys_t = buf("att_dkeys_t", N * A * Tia, torch.float32)
        dwcl = buf("att_dwcl", 8 * A, torch.float32); ops.zero(dwcl)
        ops.attention_post_bwd(N, S, Ti, Tia, A, self.KW, lengths, keys_t, q, al, de, tsh["wcl"], (self.flat_p, ov), dkeys_t,
                               (g, ov), dwcl)
        dkeys = buf("att_dkeys", N * Pi * A, torch.float32); ops.zero(dkeys)
        ops.keys_transpose_add(dkeys, dkeys_t, N, Ti, Tia, Pi, self.padl, A)
        dkeys_T = buf("att_dkeys_T", N * Pi * A, T_)
        ops.copy3d(dkeys, dkeys_T, 1, N * Pi, A, (0, A), (0, A))
        om = o("attention_decoder/memory_layer/kernel")
        ops.gemm(enc.buf, dkeys_T, g, E, A, N * Pi, E, A, A, a_mode=1, b_mode=1, c_off=om, accumulate=2, split_k=sk(N * Pi, E, A))
        ops.gemm(dkeys_T, W, enc.grad, N * Pi, E, A, A, A, E, a_mode=0, b_mode=0, b_off=om, accumulate=1)
        # dvalues[n] += align[n]^T . dctx[n], all utterances in one batched launch
        ops.gemm(al_t, dctx_t, enc.grad, Ti, E, S1, Tia, E, E, a_mode=1, b_mode=1, c_off=self.padl * E, accumulate=1,
                 batch=N, batch_strides=(S1 * Tia, S1 * E, Pi * E))

