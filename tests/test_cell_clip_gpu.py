"""hparam lstm_cell_clip: tf.contrib.rnn.LSTMBlockCell's cell_clip as a switch.  The reference builds its
cells without the argument (modules.py:41-42, tacotron2.py:69-70) and this build reads TF 1.7's default as "no clipping" -
an assumption nothing in this container can check (oracle/taco2_oracle.py: lstm_block_cell), and one that matters: the
expand BiLSTM's cell state reaches |c| = 3.4 at the benchmark shape, beyond the 3 an older TF default would clip at.  With
the switch a maintainer with TensorFlow settles it in one run.  Here: every LSTM kernel family honours the value -
step launches (exact fp32), the persistent BiLSTM clusters, the wide decoder kernels, the attention clusters, the
one-launch synthesis loop and the packed step products - against the oracle's cell with the same clip (forward clip,
straight-through gradient, as the fused TF op)."""
import numpy as np
import pytest
import torch

from util import check_flips, make_batch, oracle_report, small_hparams, stabilise_targets

pytestmark = pytest.mark.gpu


def _max_cell(m, names):
    return {k: float(m._bufs[k].abs().max().item()) for k in names if k in m._bufs}


def test_step_kernels_clip_like_the_oracle(dev):
    """Small widths, exact fp32 (launch-per-step LSTM kernels everywhere), clip 0.15: active in the four BiLSTM
    cells (a tighter clip pins every state, the decoder outputs become nearly constant and BatchNorm in training mode
    amplifies their fp32 rounding noise to 1e-3 - measured; the attention and decoder cells' clip is exercised by the
    shipped-width and synthesis tests below).  Outputs
    against the oracle's free pass, loss and every gradient on the GPU pass' own ReLU branches (util.oracle_report: with
    every state pinned to +-0.005 many pre-activations sit within rounding of zero)."""
    from test_taco2_fullwidth_gpu import BOUNDS
    from nspeech_amd.models import create_model
    hp = small_hparams(lstm_cell_clip=0.15)
    m = create_model("taco2", hp, device="cuda:0", dtype="fp32", seed=3)
    N, Ti, To = 3, 12, 20
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=11)
    mel, lin = stabilise_targets(hp, m.numpy_params(), m.numpy_stats(), inputs, lengths, mel, lin)
    rep = oracle_report(m, hp, inputs, lengths, mel, lin)          # outputs against the oracle's FREE pass
    cells = _max_cell(m, ("dec_c1", "dec_c2", "dec_ca", "expl_c_fw", "expl_c_bw", "encl_c_fw", "encl_c_bw"))
    # no state beyond the clip anywhere, and the BiLSTM cells sit AT it
    assert len(cells) == 7 and all(v <= 0.15 + 1e-7 for v in cells.values()), cells
    assert sum(abs(v - 0.15) < 1e-7 for v in cells.values()) >= 4, cells
    assert all(rep["paths"][k] == "step" for k in rep["paths"] if not k.startswith("attn")), rep["paths"]   # the LSTM step kernels
    b = BOUNDS["fp32"]
    for k, (l2, mx, l1) in rep["out"].items():
        assert mx < 5e-4, (k, l2, mx, l1)
    got, want = rep["loss"]
    assert abs(got - want) < 1e-5 * abs(want)
    bad = [(k, v) for k, v in rep["grad"].items() if not (v[0] < 2e-3 and v[1] < 4e-3)]
    assert not bad, bad[:6]
    # and the clip changes the function: the same batch without it gives other outputs
    mel_clip = m.mel_outputs.float().clone()
    hp0 = small_hparams()
    m0 = create_model("taco2", hp0, device="cuda:0", dtype="fp32", seed=3)
    m0.initialize(inputs, lengths, None, mel, lin)
    assert (m0.mel_outputs - mel_clip).abs().max().item() > 1e-3


@pytest.mark.parametrize("mode", ["bf16x3", "mixed"])
def test_persistent_kernels_clip_at_three_where_the_state_goes_beyond(dev, mode):
    """Shipped widths, clip 3.0, and weights that drive the cell states past 3 (input / cell-input / forget biases of the
    expand BiLSTM and of the second decoder LSTM raised to +3: the state then grows by ~0.9 per step) - the situation of
    the benchmark's 1000-step expand BiLSTM at |c| = 3.4, at a test-sized shape.  All persistent kernels (asserted), every
    output and gradient against the oracle within the bounds of tests/test_taco2_fullwidth_gpu.py."""
    from test_taco2_fullwidth_gpu import BOUNDS, PATHS
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    hp = hparams_mod.load("taco2")
    hp.lstm_cell_clip = 3.0
    N, Ti, To = 4, 32, 50
    m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=5)
    p = m.numpy_params()
    for name, H in (("expand/encoder_lstm/fw/lstm_cell/bias", hp.expand_lstm_units),
                    ("expand/encoder_lstm/bw/lstm_cell/bias", hp.expand_lstm_units),
                    ("decoder/lstm_2/bias", hp.decoder_lstm_units)):
        key = [k for k in p if k.endswith(name)]
        assert len(key) == 1, (name, [k for k in p if "bias" in k][:40])
        b = p[key[0]].copy()
        b[:H] += 3.0; b[H:2 * H] += 3.0; b[2 * H:3 * H] += 3.0          # gate order i, j, f, o
        p[key[0]] = b
    m.load_numpy(p, m.numpy_stats())
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=N + 20)
    rep = oracle_report(m, hp, inputs, lengths, mel, lin, stabilise=2e-3)
    m.check_status()
    for k, v in PATHS[mode].items():
        assert rep["paths"].get(k) == v, (k, rep["paths"])
    cells = _max_cell(m, ("dec_c2", "expl_c_fw", "expl_c_bw"))
    assert all(abs(v - 3.0) < 1e-5 for v in cells.values()), cells           # clipped AT 3: the state did try to go beyond
    b = BOUNDS[mode]
    check_flips(rep, mode)
    for k, (l2, mx, l1) in rep["out"].items():
        assert mx < b["out"], (k, l2, mx, l1)
    got, want = rep["loss"]
    assert abs(got - want) < b["loss"] * abs(want), rep["loss"]
    bad = [(k, v) for k, v in rep["grad"].items() if not (v[0] < b["grad_l2"] and v[1] < b["grad_max"])]
    assert not bad, bad[:6]


@pytest.mark.parametrize("N,mode", [(2, "fp32"), (2, "mixed"), (5, "mixed"), (3, "fp32")])
def test_synthesis_paths_clip_like_the_oracle(dev, N, mode):
    """Free-running synthesis at the shipped widths with clip 0.01: the one-launch decoder loop (N <= 2), the packed step
    products (`mixed`, N > 2) and the step launches (fp32, N > 2)."""
    from test_infer_gpu import _oracle_infer
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    hp = hparams_mod.load("taco2")
    hp.max_iters = 24
    hp.lstm_cell_clip = 0.01
    m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=4)
    ti, tl, tm, tn = make_batch(hp, 2, 20, 30, seed=1)
    m.add_optimizer(0)
    m.step(ti, tl, tm, tn)
    inputs, lengths, _, _ = make_batch(hp, N, 33, 10, seed=6)
    out = _oracle_infer(hp, m, inputs, lengths)
    m.use_graph = False
    m.initialize(inputs, lengths)
    m.check_status()
    assert m.last_paths["decode"] == ("persistent" if N <= 2 else "rows32" if mode == "mixed" else "step")
    for name in ("decoder_outputs", "mel_outputs", "alignments"):
        got, ref = getattr(m, name).float().cpu().numpy(), out[name].numpy()
        err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
        assert err < 5e-5, (name, err)
    hp.lstm_cell_clip = 0.0
    ref0 = _oracle_infer(hp, m, inputs, lengths)
    assert np.abs(ref0["mel_outputs"].numpy() - out["mel_outputs"].numpy()).max() > 1e-4          # the clip was active


def test_cell_clip_with_zoneout_is_refused(dev):
    from nspeech_amd.models import create_model
    hp = small_hparams(lstm_cell_clip=1.0, zoneout_rate=0.1)
    with pytest.raises(ValueError):
        create_model("taco2", hp, device="cuda:0", dtype="fp32", seed=3)
