"""The whole-chip persistent recurrences beside a FOREIGN kernel (under data parallelism RCCL's
channel kernels - a few dozen 256..512-thread workgroups that sit on their CUs until the peers have answered - share
the chip with attn_cluster_* / lstm_wide_* / lstm_cluster2_*, whose grids need every workgroup resident at once).

One-GPU proxy: ns_occupy puts RCCL-shaped workgroups on the chip from a second stream and a one-thread wait kernel on
the main stream holds the kernel under test back until they are resident, so the persistent kernel is DISPATCHED onto a
chip that has no room for all of its workgroups.  Required: no time-out (status words 0), results bit-equal to the
undisturbed pass; reported: the stall, which is what the overlap policy of DESIGN 7 is decided on."""
import ctypes as C

import pytest
import torch

from util import make_batch

pytestmark = pytest.mark.gpu

EXACT = ("d_energy", "d_q", "d_ga", "d_p2", "d_f1", "d_ctx_t", "d_hc", "d_h1", "d_h2", "d_keys_t", "d_enc_a", "d_enc_b",
         "d_act_a", "d_act_b", "d_mel", "mel_out", "lin_out", "dec_h1", "dec_h2", "dec_hc", "dec_al", "expl_h", "encl_h")


class _Occupier(object):
    def __init__(self, dev, blocks, threads, lds, heavy, usec):
        from nspeech_amd import _lib as L
        self.L, self.lib = L, L.lib()
        self.side = torch.cuda.Stream(device=dev)
        self.counter = torch.zeros(4, dtype=torch.int32, device=dev)
        self.shape = (blocks, threads, lds, heavy, usec)
        self.on = False
        self.times = []        # (name, start event, end event)

    def before(self):
        if not self.on:
            return
        blocks, threads, lds, heavy, usec = self.shape
        main = torch.cuda.current_stream()
        self.counter.zero_()
        ev = torch.cuda.Event()
        ev.record(main)
        self.side.wait_event(ev)
        self.L.check(self.lib.ns_occupy(blocks, threads, lds, heavy, C.c_double(usec), C.c_void_p(self.counter.data_ptr()),
                                        C.c_void_p(self.side.cuda_stream)), "ns_occupy")
        self.L.check(self.lib.ns_wait_counter(C.c_void_p(self.counter.data_ptr()), blocks, C.c_double(2e5),
                                              C.c_void_p(main.cuda_stream)), "ns_wait_counter")

    def wrap(self, name, fn):
        def call(*a, **kw):
            self.before()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **kw)
            e1.record()
            self.times.append((name + ":" + str(a[0]), e0, e1))
            return r
        return call


# (workgroups, threads, LDS bytes, heavy = 128 VGPRs per lane)
@pytest.mark.parametrize("shape", [(32, 512, 65536, 1), (64, 256, 32768, 1), (32, 512, 65536, 0)])
def test_persistent_recurrences_survive_a_foreign_kernel(dev, monkeypatch, shape):
    from nspeech_amd import hparams as hparams_mod, ops
    from nspeech_amd.models import create_model
    hp = hparams_mod.load("taco2")
    N, Ti, To = 32, 48, 100
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=3)
    m = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=5)
    m.overlap_wgrads = False
    usec = 800.0
    occ = _Occupier(dev, shape[0], shape[1], shape[2], shape[3], usec)
    for name in ("lstm_wide", "lstm_cluster", "taco2_attn_cluster"):
        monkeypatch.setattr(ops, name, occ.wrap(name, getattr(ops, name)))

    def one_pass():
        occ.times = []
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
        torch.cuda.synchronize()
        m.check_status()
        snap = {k: m._bufs[k].clone() for k in EXACT if k in m._bufs}
        return snap, m.flat_g.clone(), [(n, a.elapsed_time(b)) for n, a, b in occ.times]

    one_pass()                                   # warm-up (allocations, first-touch)
    ref, gref, t_ref = one_pass()
    assert len(ref) >= 18
    assert m.last_paths["attn:fwd"] == "cluster" and m.last_paths["dec1:bwd"] == "wide" and m.last_paths["expl:bwd"] == "cluster"
    occ.on = True
    got, g, t_occ = one_pass()
    for k, v in got.items():
        assert torch.equal(v, ref[k]), (k, (v.float() - ref[k].float()).abs().max().item())
    assert (g - gref).abs().max().item() <= 2e-5 * gref.abs().max().item()
    assert len(t_ref) == len(t_occ) >= 8
    print("\nforeign kernel: %d workgroups x %d threads x %d B LDS, heavy %d, for %.0f us" % (shape + (usec,)))
    worst = 0.0
    for (n, a), (_, b) in zip(t_ref, t_occ):
        print("  %-28s alone %7.3f ms   beside the occupier %7.3f ms   stall %+7.3f ms" % (n, a, b, b - a))
        worst = max(worst, b - a)
    # a persistent kernel cannot finish before its last workgroup is placed, i.e. before the occupier leaves; it must
    # not take much longer than that either (no livelock, no time-out): stall <= the occupier's life + slack
    assert worst < usec / 1e3 + 0.6, worst
