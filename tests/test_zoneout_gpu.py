"""The zoneout option of the two decoder LSTMs (north_star: "2-layer Zoneout-LSTM decoder").  The
reference builds plain LSTMBlockCells (tacotron2.py:69-70), so the shipped rate is 0 and every other parity test runs the
plain path; here rate 0.1 is compared with the oracle's zoneout_cell() on the SAME masks - the kernels draw them from a
counter-based generator that oracle/taco2_oracle.py: zoneout_masks restates in NumPy integer arithmetic - forward and
backward, through the step kernels (small widths, exact fp32) and the persistent wide-cell kernels (shipped widths)."""
import numpy as np
import pytest
import torch

from util import check_flips, make_batch, oracle_report, small_hparams, stabilise_targets

pytestmark = pytest.mark.gpu


def _with_zoneout(m, S, N, H):
    """The oracle-side description of model m's masks for its current global step."""
    from oracle import taco2_oracle as O
    masks = {}
    for layer in (1, 2):
        thr_c, thr_h, seed_c, seed_h = m.zoneout_args(layer)
        masks[layer] = (O.zoneout_masks(seed_c, thr_c, S, N, H), O.zoneout_masks(seed_h, thr_h, S, N, H))
    return dict(rate=m.zoneout_rate, masks=masks)


def test_rate_zero_is_the_plain_cell(dev):
    """zoneout_rate 0 hands the kernels no zoneout block at all: the reference path, unchanged."""
    from nspeech_amd.models import create_model
    hp = small_hparams()
    m = create_model("taco2", hp, device="cuda:0", dtype="fp32", seed=3)
    assert m.zoneout_rate == 0.0 and m.zoneout_args(1) is None and m.zoneout_args(2) is None


@pytest.mark.parametrize("shape", [(3, 11, 20), (4, 16, 40)])
def test_zoneout_training_step_kernels_match_oracle(dev, shape):
    from oracle import taco2_oracle as O
    from nspeech_amd.models import create_model
    N, Ti, To = shape
    hp = small_hparams(zoneout_rate=0.1)
    m = create_model("taco2", hp, device="cuda:0", dtype="fp32", seed=3)
    S = To // hp.outputs_per_step
    O.ZONEOUT = _with_zoneout(m, S, N, hp.decoder_lstm_units)
    try:
        kept = [float(k.mean()) for pair in O.ZONEOUT["masks"].values() for k in pair]
        assert all(0.03 < k < 0.2 for k in kept), kept               # the masks are neither empty nor everything
        inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=N)
        mel, lin = stabilise_targets(hp, m.numpy_params(), m.numpy_stats(), inputs, lengths, mel, lin)
        rep = oracle_report(m, hp, inputs, lengths, mel, lin)
    finally:
        O.ZONEOUT = None
    assert rep["paths"]["dec1:fwd"] == "step" and rep["paths"]["dec2:bwd"] == "step", rep["paths"]
    check_flips(rep, "fp32")
    for k, (l2, mx, l1) in rep["out"].items():
        assert mx < 5e-4, (k, mx)
    assert abs(rep["loss"][0] - rep["loss"][1]) < 1e-5 * abs(rep["loss"][1])
    bad = [(k, v) for k, v in rep["grad"].items() if not v[1] < 2e-3]
    assert not bad, bad
    # and the masks matter: the same batch through plain cells gives other decoder outputs
    plain = create_model("taco2", small_hparams(), device="cuda:0", dtype="fp32", seed=3)
    plain.initialize(inputs, lengths, None, mel, lin)
    d = (plain.decoder_outputs.float() - m.decoder_outputs.float()).abs().max().item()
    assert d > 1e-3, d


@pytest.mark.parametrize("mode", ["bf16x3", "mixed"])
def test_zoneout_wide_kernels_match_oracle_at_shipped_widths(dev, mode):
    """lstm_wide_fwd_kernel / lstm_wide_bwd_ps_kernel apply the same masks (and their gradient) inside the persistent
    recurrences; bounds as tests/test_taco2_fullwidth_gpu.py."""
    from oracle import taco2_oracle as O
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    from test_taco2_fullwidth_gpu import BOUNDS
    hp = hparams_mod.load("taco2")
    hp.zoneout_rate = 0.1
    N, Ti, To = 3, 24, 40
    m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=5)
    O.ZONEOUT = _with_zoneout(m, To // hp.outputs_per_step, N, hp.decoder_lstm_units)
    try:
        inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=N + 20)
        mel, lin = stabilise_targets(hp, m.numpy_params(), m.numpy_stats(), inputs, lengths, mel, lin)
        rep = oracle_report(m, hp, inputs, lengths, mel, lin)
    finally:
        O.ZONEOUT = None
    m.check_status()
    assert rep["paths"]["dec1:fwd"] == "wide" and rep["paths"]["dec2:fwd"] == "wide", rep["paths"]
    if mode == "mixed":
        assert rep["paths"]["dec1:bwd"] == "wide" and rep["paths"]["dec2:bwd"] == "wide", rep["paths"]
    check_flips(rep, mode)
    b = BOUNDS[mode]
    assert rep["out"]["mel_outputs"][2] < b["mel_l1"], rep["out"]["mel_outputs"]
    for k, (l2, mx, l1) in rep["out"].items():
        assert mx < b["out"], (k, l2, mx, l1)
    bad = [(k, v) for k, v in rep["grad"].items() if not (v[0] < b["grad_l2"] and v[1] < b["grad_max"])]
    assert not bad, bad
    print("zoneout %s: worst gradient rel L2 %.2e" % (mode, max(v[0] for v in rep["grad"].values())))


@pytest.mark.parametrize("mode,path", [("fp32", "step"), ("mixed", "rows32")])
def test_zoneout_inference_is_the_expectation(dev, mode, path):
    """Synthesis with a zoneout rate: c = z c_prev + (1 - z) c', h likewise, through the step launches (exact fp32: ns_lstm_step;
    `mixed`: the packed step products, ns_rows32)."""
    from oracle import taco2_oracle as O
    from nspeech_amd.models import create_model
    hp = small_hparams(max_iters=6, zoneout_rate=0.1)
    m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=2)
    inputs, lengths, _, _ = make_batch(hp, 2, 12, 10, seed=4)
    p = {k: torch.tensor(v, dtype=torch.float64) for k, v in m.numpy_params().items()}
    p.update({k: torch.tensor(v, dtype=torch.float64) for k, v in m.numpy_stats().items()})
    with torch.no_grad():
        out = O.taco2_forward(p, hp.values(), torch.tensor(inputs), torch.tensor(lengths), zoneout=dict(rate=0.1))
        plain = O.taco2_forward(p, hp.values(), torch.tensor(inputs), torch.tensor(lengths))
    m.initialize(inputs, lengths)
    assert m.last_paths["decode"] == path
    for name in ("decoder_outputs", "mel_outputs", "alignments"):
        got = getattr(m, name).float().cpu().numpy()
        ref = out[name].numpy()
        assert np.abs(got - ref).max() < 5e-4 * max(1.0, np.abs(ref).max()), name
    ref, pl = out["decoder_outputs"].numpy(), plain["decoder_outputs"].numpy()
    assert np.abs(ref - pl).max() > 0.05 * np.abs(pl).max()        # and it is not the plain cell (measured: 0.2)
