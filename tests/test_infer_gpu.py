"""Free-running synthesis (TacoTestHelper feedback, inference BatchNorm) against the oracle, and
the Synthesizer surface end to end (text -> ids -> mel/linear -> Griffin-Lim wav)."""
import numpy as np
import pytest
import torch

from util import make_batch, small_hparams

pytestmark = pytest.mark.gpu


def _oracle_infer(hp, m, inputs, lengths):
    from oracle import taco2_oracle as O
    p = {k: torch.tensor(v, dtype=torch.float64) for k, v in m.numpy_params().items()}
    p.update({k: torch.tensor(v, dtype=torch.float64) for k, v in m.numpy_stats().items()})
    with torch.no_grad():
        return O.taco2_forward(p, hp.values(), torch.tensor(inputs), torch.tensor(lengths))


@pytest.mark.parametrize("trained", [False, True])
def test_inference_matches_oracle(dev, trained):
    from nspeech_amd.models import create_model
    hp = small_hparams(max_iters=6)
    m = create_model("taco2", hp, device="cuda:0", dtype="fp32", seed=2)
    if trained:   # one optimiser step so that BN moving statistics and weights are non-trivial
        inputs, lengths, mel, lin = make_batch(hp, 3, 9, 15, seed=1)
        m.add_optimizer(0)
        m.step(inputs, lengths, mel, lin)
    inputs, lengths, _, _ = make_batch(hp, 2, 12, 10, seed=4)
    out = _oracle_infer(hp, m, inputs, lengths)
    m.initialize(inputs, lengths)
    m.check_status()
    assert m.last_paths["decode"] == "persistent"          # two utterances: the one-launch decoder loop (ns_taco2_decode)
    assert tuple(m.mel_outputs.shape) == (2, 6 * hp.outputs_per_step, hp.num_mels)
    assert tuple(m.alignments.shape) == (2, 12, 6)
    for name in ("decoder_outputs", "mel_outputs", "linear_outputs", "alignments"):
        got = getattr(m, name).float().cpu().numpy()
        ref = out[name].numpy()
        assert np.abs(got - ref).max() < 5e-4 * max(1.0, np.abs(ref).max()), name


@pytest.mark.parametrize("N", [3, 2, 1])          # 3: the launch-per-step loop; 2, 1: the persistent decoder loop
def test_inference_graph_replay_matches_eager(dev, N):
    """Second call with one signature captures a HIP graph, later calls replay it: new inputs and lengths (same
    shapes) must give what a fresh eager pass gives, in the mode the bench uses."""
    from nspeech_amd.models import create_model
    hp = small_hparams(max_iters=5)
    m = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=3)
    e = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=3)
    e.use_graph = False
    for seed in (1, 2, 3, 4):           # eager, capture + replay, replay, replay
        inputs, lengths, _, _ = make_batch(hp, N, 11, 10, seed=seed)
        m.initialize(inputs, lengths)
        e.initialize(inputs, lengths)
        assert (m._infer_graph["graph"] is not None) == (seed > 1)
        for name in ("mel_outputs", "linear_outputs", "alignments"):
            got, ref = getattr(m, name).float(), getattr(e, name).float()
            assert torch.equal(got, ref), (seed, name, (got - ref).abs().max().item())
    assert m.last_paths["decode"] == ("persistent" if N <= 2 else "rows32")


@pytest.mark.parametrize("N,mode", [(1, "mixed"), (2, "mixed"), (1, "fp32"), (1, "bf16")])
def test_persistent_decoder_loop_matches_the_step_launches_at_shipped_widths(dev, N, mode):
    """ns_taco2_decode (attention clusters + register-resident decoder LSTMs + folded frame feedback in ONE launch) against
    the launch-per-step loop on the same weights at the shipped widths (attention 256, LSTM(1024), 512-wide memory):
    40 free-running steps.  The two differ in arithmetic (exact fp32 FMAs against three split-bf16 passes in `mixed`,
    fp32 master weights against bf16 ones in `bf16`), and a free-running loop feeds its own rounding back."""
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    hp = hparams_mod.load("taco2")
    hp.max_iters = 40
    inputs, lengths, _, _ = make_batch(hp, N, 37, 10, seed=5)
    outs = []
    for use in (True, False):
        m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=4)
        m.use_decode_kernel = use
        m.use_rows32 = False
        m.use_graph = False
        m.initialize(inputs, lengths)
        torch.cuda.synchronize()
        m.check_status()
        assert m.last_paths["decode"] == ("persistent" if use else "step")
        outs.append({k: getattr(m, k).float().clone() for k in ("decoder_outputs", "mel_outputs", "alignments")})
        del m
    tol = {"mixed": 2e-5, "fp32": 2e-5, "bf16": 1e-3}[mode]      # measured 7e-7, 6e-7, 1.4e-5
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        err = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
        print("%s N %d %s: persistent vs step launches max %.3e" % (mode, N, k, err))
        assert err < tol, (k, err)
    al = outs[0]["alignments"]
    assert (al.sum(1) - 1.0).abs().max().item() < 1e-4


@pytest.mark.parametrize("N", [5, 32])
def test_packed_step_products_match_the_step_launches_at_shipped_widths(dev, N):
    """The batched free-running loop on packed weights (ns_rows32: folded frame feedback, context term of the prenet from
    the projected memory, seven launches a step) against the launch-per-step loop of round 2 on the same weights, `mixed`,
    shipped widths, 40 free-running steps, a full batch of 32 and a ragged one."""
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    hp = hparams_mod.load("taco2")
    hp.max_iters = 40
    inputs, lengths, _, _ = make_batch(hp, N, 37, 10, seed=5)
    outs = []
    for use in (True, False):
        m = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=4)
        m.use_rows32 = use
        m.use_graph = False
        m.initialize(inputs, lengths)
        torch.cuda.synchronize()
        assert m.last_paths["decode"] == ("rows32" if use else "step")
        outs.append({k: getattr(m, k).float().clone() for k in ("decoder_outputs", "mel_outputs", "alignments")})
        del m
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        err = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
        print("N %d %s: packed step products vs step launches max %.3e" % (N, k, err))
        assert err < 2e-5, (k, err)
    assert (outs[0]["alignments"].sum(1) - 1.0).abs().max().item() < 1e-4


# 1, 2: taco2_decode_kernel (one launch); 3, 32: the launch-per-step loop (fp32) / the packed step products (mixed)
@pytest.mark.parametrize("N", [1, 2, 3, 32])
@pytest.mark.parametrize("mode", ["fp32", "mixed"])
def test_free_running_decode_matches_oracle_at_shipped_widths(dev, N, mode):
    """free-running synthesis at the SHIPPED widths (attention 256, LSTM(1024), 512-wide memory, 1025
    bins) against the float64 oracle - 40 decoder steps that feed their own last frame back (helpers.py:7-38), then the
    postnet, the expand net and the linear head with moving-average BatchNorm.  The model takes one optimiser step first so
    that the moving statistics and the biases are not their initial values."""
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    hp = hparams_mod.load("taco2")
    hp.max_iters = 40
    m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=4)
    ti, tl, tm, tn = make_batch(hp, 2, 20, 30, seed=1)
    m.add_optimizer(0)
    m.step(ti, tl, tm, tn)
    inputs, lengths, _, _ = make_batch(hp, N, 37, 10, seed=5)
    out = _oracle_infer(hp, m, inputs, lengths)
    m.use_graph = False
    m.initialize(inputs, lengths)
    m.check_status()
    assert m.last_paths["decode"] == ("persistent" if N <= 2 else "rows32" if mode == "mixed" else "step")
    # measured (profiles/r04_parity_fullwidth.txt, "free-running") -> bound; a free-running loop feeds its rounding back
    # fp32: 7.5e-7 rel max (mel); mixed: 9.6e-7 (mel), linear_outputs 6.4e-5 (the bf16 expand net)
    tol = {"fp32": dict(rel=2e-5, mel_l1=5e-6), "mixed": dict(rel=2e-5, mel_l1=5e-6)}[mode]
    for name in ("decoder_outputs", "mel_outputs", "alignments", "linear_outputs"):
        got = getattr(m, name).float().cpu().numpy()
        ref = out[name].numpy()
        err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
        l1 = np.abs(got - ref).mean()
        print("free-running %s N %d %s: rel max %.3e, L1 %.3e" % (mode, N, name, err, l1))
        if name == "linear_outputs" and mode == "mixed":
            assert err < 2e-3, (name, err)        # the bf16 expand net
        else:
            assert err < tol["rel"], (name, err)
        if name == "mel_outputs":
            assert l1 < tol["mel_l1"], l1          # north_star: mel within 1e-3 L1


def test_synthesizer_end_to_end(dev):
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.synthesizer import Synthesizer
    hp = small_hparams(max_iters=8, num_freq=1025, num_mels=80)
    hparams_mod.set_hparams(hp)
    synth = Synthesizer(hp, dtype="bf16").load(None, "taco2")
    wav, mel, lin = synth.synthesize("Hello, World.")
    T = 8 * hp.outputs_per_step
    assert mel.shape == (T, 80) and lin.shape == (T, 1025)
    assert wav.ndim == 1 and len(wav) <= (T - 1) * 250 + 1000 and np.isfinite(wav).all()


def test_config3_cmudict_longform_decode(dev):
    """BASELINE config 3: Tacotron-2 at the shipped widths on `{ARPAbet}` phoneme input (what CMUDict substitution
    produces), max_iters=400 -> 2000 mel frames = 25 s of audio through the free-running decoder, the postnet, the
    expand net and the 60-iteration Griffin-Lim.  No oracle finishes this size in seconds, so: ids are the golden
    ARPAbet ids, shapes and lengths follow SURVEY Q7, every output is finite, the alignment rows are distributions,
    and a second call (HIP-graph replay) reproduces the first."""
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.synthesizer import Synthesizer
    from nspeech_amd.utils.text import text_to_sequence
    hp = hparams_mod.load("taco2")
    hp.max_iters = 400
    hparams_mod.set_hparams(hp)
    text = "Turn left on {HH AW1 S S T AH0 N} Street, then {R AY1 T} at the {L AY1 T}."
    ids = text_to_sequence(text, ["english_cleaners"])
    assert [107, 83, 132, 132, 134, 74, 120] == ids[13:20] and max(ids) < 149 and ids[-1] == 1   # SURVEY 8c golden ids
    synth = Synthesizer(hp, dtype="mixed").load(None, "taco2")
    outs = [synth.synthesize(text) for _ in range(3)]          # eager pass, graph capture, replay
    wav, mel, lin = outs[0]
    T = 400 * hp.outputs_per_step
    assert mel.shape == (T, hp.num_mels) and lin.shape == (T, hp.num_freq)
    assert np.isfinite(mel).all() and np.isfinite(lin).all() and np.isfinite(wav).all()
    assert wav.ndim == 1 and len(wav) <= (T - 1) * 250 + 1000
    al = synth.model.alignments[0].float().cpu().numpy()       # [T_in, steps]
    assert al.shape == (len(ids), 400) and np.abs(al.sum(axis=0) - 1.0).max() < 1e-4 and (al >= 0).all()
    for w2, m2, l2 in outs[1:]:
        assert np.array_equal(m2, mel) and np.array_equal(l2, lin)
