"""Tacotron-1 at the SHIPPED widths of hparams/taco1.yaml (16-bank / 8-bank CBHG, 128-unit BiGRUs, 256-unit attention and
decoder GRUs, 80 mel / 1025 linear bins - BASELINE config 1's model) against the float64 oracle: outputs, losses and
EVERY gradient tensor, per precision mode, with the persistent GRU kernels (csrc/gru.hip) asserted through
model.last_paths.  The small-width tests of test_taco1_gpu.py run the launch-per-step kernels (H = 64).

As in test_taco2_fullwidth_gpu.py the gradients are compared with the oracle on the GPU pass' own ReLU branches
(taco2_oracle.MASK_FORCE: ~1e6 pre-activations per pass always put a few within rounding of zero and one flipped kink
moves whole gradient columns), the OUTPUTS with the oracle's free pass, and the branch differences are bounded in count
and in distance from the kink (util.check_flips).

Reference semantics: tacotron.py:38-107, modules.py:109-191 (CBHG, highway, BiGRU), rnn_wrappers.py:25-31."""
import numpy as np
import pytest
import torch

from util import check_flips, make_batch, rel_l2, rel_max

pytestmark = pytest.mark.gpu

# measured (profiles/r05_taco1_parity.txt) -> bound; the modes as in test_taco2_fullwidth_gpu.py.  Tacotron-1 has no bf16
# sub-network in `mixed`: every forward product runs in three split-bf16 passes, the backward pass in one.
BOUNDS = {
    "fp32": dict(out=3e-4, mel_l1=1e-4, grad_l2=3e-4, grad_l2_median=1e-4, grad_max=6e-4, loss=1e-5),
    "bf16x3": dict(out=3e-4, mel_l1=1e-4, grad_l2=4e-4, grad_l2_median=2e-4, grad_max=8e-4, loss=1e-5),
    "mixed": dict(out=3e-4, mel_l1=1e-4, grad_l2=6e-2, grad_l2_median=1.5e-2, grad_max=0.3, loss=1e-4),
    "bf16": dict(out=0.12, mel_l1=3e-2, grad_l2=0.6, grad_l2_median=0.12, grad_max=0.6, loss=2e-3),
}
FLIPS = {       # family: (largest |x| / rms at a differing branch, largest fraction of differing branches)
    "fp32": dict(pre=(1e-4, 5e-5), bank=(1e-4, 5e-5), hw=(1e-4, 5e-5)),
    "bf16x3": dict(pre=(2.5e-4, 1e-4), bank=(2.5e-4, 1e-4), hw=(2.5e-4, 1e-4)),
    "mixed": dict(pre=(2.5e-4, 1e-4), bank=(2.5e-4, 1e-4), hw=(2.5e-4, 1e-4)),
    "bf16": dict(pre=(8e-2, 1e-2), bank=(0.3, 2e-2), hw=(0.3, 2e-2)),
}
SEQ = {"enc_gru:fwd": "seq", "enc_gru:bwd": "seq", "post_gru:fwd": "seq", "post_gru:bwd": "seq", "gru_1:fwd": "seq",
       "gru_1:bwd": "seq", "gru_2:fwd": "seq", "gru_2:bwd": "seq"}
ATT = {"attn:fwd": "cluster", "attn:bwd": "cluster"}          # csrc/attn_gru.hip: exact fp32 products, every mode
PATHS = {"fp32": dict({k: "step" for k in SEQ}, **ATT), "bf16x3": dict(SEQ, **ATT), "mixed": dict(SEQ, **ATT),
         "bf16": dict(SEQ, **ATT)}


def _families(hp, S):
    cb = lambda K, nproj: ["bank"] * (K + nproj - 1) + ["hw"] * 4
    return (["pre"] * 2 + cb(hp.encoder_cbhg_banks, len(hp.encoder_cbhg_bank_sizes)) + ["pre"] * (2 * S) +
            cb(hp.post_cbhg_banks, len(hp.post_cbhg_bank_sizes) + 1))


def _model_masks(m):
    """The ReLU branch masks of a Tacotron (taco1) forward_train() from its buffers, in the oracle's call order: encoder
    prenet (2), encoder CBHG (banks, relu projections, 4 highway H), decoder prenet (2 per step), post CBHG."""
    hp, d, B = m._hparams, m.dims, m._bufs
    N, Ti, To, S, Pi, Po = d["N"], d["Ti"], d["To"], d["S"], d["Pi"], d["Po"]
    pl = m.padl

    def act(name, P, T, C):
        return (B[name][:N * P * C].float().view(N, P, C)[:, pl:pl + T] > 0).cpu().numpy()

    out = [act("a:pre1", Pi, Ti, hp.encoder_prenet[0]), act("a:pre2", Pi, Ti, hp.encoder_prenet[1])]

    def cbhg(name, P, T, K, proj, widths=(128, 128, 128, 128)):
        o = [act("c:%s_b%d_z" % (name, k), P, T, 128) for k in range(1, K + 1)]
        o += [act("c:%s_p%d_z" % (name, i + 1), P, T, size) for i, size in enumerate(proj[:-1])]
        o += [act("a:%s_hw%d_h" % (name, i), P, T, widths[i]) for i in range(4)]
        return o
    # with a speaker embedding every encoder highway layer reads [h | projection]: the width doubles per layer
    ew = (256, 512, 1024, 2048) if m.Dsp else (128, 128, 128, 128)
    out += cbhg("enc", Pi, Ti, hp.encoder_cbhg_banks, list(hp.encoder_cbhg_bank_sizes), ew)
    A = hp.attention_dim
    XA = 128 + m.Dsp + A
    p1 = B["dec_p1"][:N * (S + 1) * 256].float().view(N, S + 1, 256)
    p2 = B["dec_xa"][:N * (S + 1) * XA].float().view(N, S + 1, XA)[:, :, :128]
    for s in range(1, S + 1):
        out.append((p1[:, s] > 0).cpu().numpy())
        out.append((p2[:, s] > 0).cpu().numpy())
    out += cbhg("post", Po, To, hp.post_cbhg_banks, list(hp.post_cbhg_bank_sizes) + [hp.num_mels])
    return out


def _oracle(hp, params, stats, inputs, lengths, mel, lin, need_grad=True, force=None, log=False, spk=None):
    from oracle import taco1_oracle as O1, taco2_oracle as O2
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=need_grad) for k, v in params.items()}
    p.update({k: torch.tensor(v, dtype=torch.float64) for k, v in stats.items()})
    O2.MASK_FORCE = None if force is None else [np.asarray(x) for x in force]
    O2.MASK_LOG = [] if log else None
    O2.FLIP_LOG = [] if force is not None else None
    try:
        with torch.set_grad_enabled(need_grad):
            out = O1.taco1_forward(p, hp.values(), torch.tensor(inputs), torch.tensor(lengths),
                                   torch.tensor(mel, dtype=torch.float64), torch.tensor(lin, dtype=torch.float64),
                                   speaker_ids=None if spk is None else torch.tensor(spk))
            loss, ml, ll = O1.taco1_loss(hp.values(), out, torch.tensor(mel, dtype=torch.float64),
                                         torch.tensor(lin, dtype=torch.float64))
        masks, flips = O2.MASK_LOG, O2.FLIP_LOG
    finally:
        O2.MASK_FORCE = O2.MASK_LOG = O2.FLIP_LOG = None
    grads = None
    if need_grad:
        loss.backward()
        grads = {k: (p[k].grad.numpy() if p[k].grad is not None else np.zeros_like(params[k])) for k in params}
    return out, float(loss.detach()), grads, masks, flips


def _report(m, hp, inputs, lengths, mel, lin, margin=2e-3, spk=None):
    params, stats = m.numpy_params(), m.numpy_stats()
    mel, lin = mel.copy(), lin.copy()
    free = None
    for _ in range(3):                              # L1 targets off the free pass' predictions (sign() gradients)
        free, _, _, _, _ = _oracle(hp, params, stats, inputs, lengths, mel, lin, need_grad=False, spk=spk)
        bm = np.abs(free["mel_outputs"].numpy() - mel) < margin
        bl = np.abs(free["linear_outputs"].numpy() - lin) < margin
        if not bm.any() and not bl.any():
            break
        mel[bm] -= 10 * margin
        lin[bl] -= 10 * margin
    m.initialize(inputs, lengths, spk, mel, lin)
    got_masks = _model_masks(m)
    out, loss, grads, _, flog = _oracle(hp, params, stats, inputs, lengths, mel, lin, force=got_masks, spk=spk)
    names = _families(hp, m.dims["S"])
    assert len(names) == len(flog) == len(got_masks), (len(names), len(flog), len(got_masks))
    fams = {}
    for fam, (n, mx, rms, tot) in zip(names, flog):
        a = fams.get(fam, (0, 0, 0.0))
        fams[fam] = (a[0] + n, a[1] + tot, max(a[2], mx / max(rms, 1e-30)))
    m.backward()
    m.read_losses()
    rep = {"out": {}, "grad": {}, "loss": (m.loss, loss), "flip_families": fams, "paths": dict(m.last_paths)}
    for k in ("mel_outputs", "linear_outputs", "alignments"):
        a, b = getattr(m, k).float().cpu().numpy(), free[k].detach().numpy()
        rep["out"][k] = (rel_l2(a, b), rel_max(a, b), float(np.abs(a - b).mean()))
    got = m.numpy_grads()
    gn = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in grads.values()))
    for k in grads:
        if k.endswith("conv1d/bias") or np.linalg.norm(grads[k]) < 1e-9 * gn:      # zero true gradient in front of BatchNorm
            rep["grad"][k] = (float(np.linalg.norm(got[k] - grads[k]) / gn), float(np.abs(got[k] - grads[k]).max() / gn))
        else:
            rep["grad"][k] = (rel_l2(got[k], grads[k]), rel_max(got[k], grads[k]))
    st = m.numpy_stats()
    rep["bn"] = max(float(np.abs(st[k] - v.numpy()).max()) for k, v in out["bn_updates"].items())
    return rep


# (32, 24, 25): two 16-row groups in every recurrence
@pytest.mark.parametrize("shape", [(4, 24, 40), (32, 24, 25)])
@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "mixed", "bf16"])
def test_taco1_shipped_widths_match_oracle(dev, mode, shape):
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    hp = hparams_mod.load("taco1")
    N, Ti, To = shape
    m = create_model("taco1", hp, device="cuda:0", dtype=mode, seed=5)
    assert m.layout.shape("encoder_cbhg/conv_bank/conv1d_16/conv1d/kernel") == (16, 128, 128)
    assert m.layout.shape("decoder/gru_2/gates/kernel") == (512, 512) and m.layout.shape("dense/kernel") == (256, 1025)
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=N + 40)
    rep = _report(m, hp, inputs, lengths, mel, lin)
    m.check_status()
    for k, v in PATHS[mode].items():
        assert rep["paths"].get(k) == v, (k, rep["paths"])
    if mode != "fp32":      # the two residual GRUs ran as a pipeline in time (Tacotron._gru_pair) where the steps allow four windows
        steps = -(-To // hp.outputs_per_step)
        assert ("gru_pair" in rep["paths"]) == (steps >= 8), (steps, rep["paths"])
    b = BOUNDS[mode]
    worst = sorted(rep["grad"].items(), key=lambda kv: -kv[1][0])[:3]
    print("\ntaco1 %s %s: ReLU flips %s; outputs (rel L2, rel max, L1) %s; BN stats %.1e; worst gradients %s; median %.2e" % (
        mode, shape, {k: (v[0], v[1], float("%.2e" % v[2])) for k, v in rep["flip_families"].items()},
        {k: tuple(float("%.2e" % x) for x in v) for k, v in rep["out"].items()}, rep["bn"],
        [(k, float("%.2e" % v[0])) for k, v in worst], float(np.median([v[0] for v in rep["grad"].values()]))))
    check_flips(rep, mode, bounds=FLIPS[mode])
    assert rep["out"]["mel_outputs"][2] < b["mel_l1"], rep["out"]["mel_outputs"]
    for k, (l2, mx, l1) in rep["out"].items():
        assert mx < b["out"], (k, l2, mx, l1)
    got, want = rep["loss"]
    assert abs(got - want) < b["loss"] * abs(want), rep["loss"]
    bad = [(k, v) for k, v in rep["grad"].items() if not (v[0] < b["grad_l2"] and v[1] < b["grad_max"])]
    assert not bad, bad
    assert float(np.median([v[0] for v in rep["grad"].values()])) < b["grad_l2_median"]
    assert rep["bn"] < (1e-4 if mode != "bf16" else 2e-2)


def test_taco1_multi_speaker_at_shipped_widths(dev):
    """num_speakers > 1 at the shipped widths (tacotron.py:41-66, modules.py:157-169, rnn_wrappers.py:28-30): the encoder
    CBHG's highway widths double per layer (256 .. 2048), the BiGRU(128) starts from the speaker projection (the persistent
    kernel's h_init, per-row lengths: the backward direction meets it at a different step per utterance) and the attention
    GRU's input row is [prenet | speaker | h] - the persistent attention clusters fold the speaker rows into their biases.
    Every output and gradient against the float64 oracle in split-bf16 arithmetic."""
    from test_taco1_gpu import _spread_speaker_path
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    hp = hparams_mod.load("taco1")
    hp.num_speakers = 3
    N, Ti, To = 4, 22, 30
    m = create_model("taco1", hp, device="cuda:0", dtype="bf16x3", seed=6)
    assert m.layout.shape("decoder/attention_gru/gates/kernel") == (128 + 128 + 256, 512)
    _spread_speaker_path(m)
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=61)
    lengths = np.asarray(lengths).copy()
    lengths[0], lengths[-1] = Ti, Ti // 2
    spk = np.array([2, 0, 2, 1], np.int32)
    rep = _report(m, hp, inputs, lengths, mel, lin, spk=spk)
    m.check_status()
    for k, v in PATHS["bf16x3"].items():
        assert rep["paths"].get(k) == v, (k, rep["paths"])
    b = BOUNDS["bf16x3"]
    for k, (l2, mx, l1) in rep["out"].items():
        assert mx < b["out"], (k, l2, mx, l1)
    got, want = rep["loss"]
    assert abs(got - want) < b["loss"] * abs(want), rep["loss"]
    bad = [(k, v) for k, v in rep["grad"].items() if not (v[0] < b["grad_l2"] and v[1] < b["grad_max"])]
    assert not bad, bad[:6]
    for k in ("speaker/speaker_embed", "encoder_cbhg/dense/kernel", "decoder/dense/kernel"):
        assert rep["grad"][k][0] < b["grad_l2"]


def test_persistent_paths_equal_the_step_launches_in_the_model(dev, monkeypatch):
    """The same model pass with the persistent kernels (GRU recurrences, attention loop) and with the launch-per-step forms: both
    compute the same products from the same operands in split-bf16 arithmetic - outputs and gradients agree to rounding."""
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    from nspeech_amd.models.tacotron import Tacotron
    hp = hparams_mod.load("taco1")
    N, Ti, To = 20, 17, 30
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=9)
    res = {}
    for seq in (True, False):
        monkeypatch.setattr(Tacotron, "use_gru_seq", seq)
        monkeypatch.setattr(Tacotron, "use_attn_cluster", seq)
        m = create_model("taco1", hp, device="cuda:0", dtype="bf16x3", seed=5)
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
        m.read_losses()
        assert m.last_paths["post_gru:fwd"] == ("seq" if seq else "step") and m.last_paths["gru_1:bwd"] == ("seq" if seq else "step")
        assert m.last_paths["attn:fwd"] == m.last_paths["attn:bwd"] == ("cluster" if seq else "step")
        res[seq] = (m.mel_outputs.float().cpu().numpy(), m.linear_outputs.float().cpu().numpy(), m.numpy_grads(), m.loss)
    assert rel_max(res[True][0], res[False][0]) < 2e-5 and rel_max(res[True][1], res[False][1]) < 2e-5
    assert abs(res[True][3] - res[False][3]) < 1e-6 * abs(res[False][3])
    # Gradients: the two passes' forward values differ in the last bits, so a few of the ~1e6 ReLU pre-activations and L1
    # residuals fall on the other side of their kink and move whole gradient columns (measured: 5.9e-3 on one conv bank
    # kernel) - the typical tensor agrees to rounding, no tensor is off by more than such flips explain.  (A conv bias in
    # front of BatchNorm has a zero true gradient: what both paths hold there is cancellation noise.)
    gn = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in res[False][2].values()))
    errs = sorted((rel_l2(res[True][2][k], res[False][2][k]), k) for k in res[True][2]
                  if not k.endswith("conv1d/bias") and np.linalg.norm(res[False][2][k]) > 1e-9 * gn)
    assert errs[len(errs) // 2][0] < 1e-4, errs[len(errs) // 2]
    assert errs[-1][0] < 3e-2, errs[-1]


def test_pipelined_residual_grus_equal_the_sequential_ones(dev):
    """Tacotron._gru_pair cuts the two residual GRUs into time windows on two streams; with NS_GRU_PIPE=0 they run one
    after the other over the whole sequence.  Same kernels, same arithmetic per step: outputs and gradients agree to the
    last bits of the products whose summation order the windows change (the batched input products of GRU_2)."""
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    from nspeech_amd.models.tacotron import Tacotron
    hp = hparams_mod.load("taco1")
    N, Ti, To = 4, 24, 60
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=3)
    res = []
    for pipe in (True, False):
        Tacotron.use_gru_pipeline = pipe
        try:
            m = create_model("taco1", hp, device="cuda:0", dtype="mixed", seed=5)
            m.initialize(inputs, lengths, None, mel, lin)
            m.backward()
            m.check_status()
            assert ("gru_pair" in m.last_paths) == pipe
            res.append((m.mel_outputs.float().cpu().numpy(), m.linear_outputs.float().cpu().numpy(), m.numpy_grads()))
        finally:
            Tacotron.use_gru_pipeline = True
    (ma, la, ga), (mb, lb, gb) = res
    assert np.abs(ma - mb).max() < 1e-5 * max(1.0, np.abs(mb).max()) and np.abs(la - lb).max() < 1e-5 * max(1.0, np.abs(lb).max())
    rel = {k: np.abs(ga[k] - gb[k]).max() / (np.abs(gb[k]).max() + 1e-12) for k in gb}
    worst = max(rel.values())
    assert float(np.median(list(rel.values()))) < 1e-4 and worst < 3e-2, sorted(rel.items(), key=lambda kv: -kv[1])[:4]
