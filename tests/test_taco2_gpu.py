"""GPU parity of the Tacotron-2 training step (forward, losses, every gradient, Adam update)
against the float64 CPU oracle, in exact-fp32 mode (tight) and bf16-MFMA mode (north_star
tolerance: mel outputs within 1e-3 mean L1)."""
import numpy as np
import pytest
import torch

from util import check_flips, make_batch, oracle_report, oracle_run, same_branch_batch, small_hparams, stabilise_targets

pytestmark = pytest.mark.gpu


def _model(hp, dtype, seed=3):
    from nspeech_amd.models import create_model
    return create_model("taco2", hp, device="cuda:0", dtype=dtype, seed=seed)


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


# the last three: a single utterance; more than 32 rows (second row tile of every recurrent kernel); text longer
# than one 64-position attention chunk
@pytest.mark.parametrize("shape", [(3, 11, 20), (2, 7, 10), (5, 20, 35), (1, 3, 20), (33, 9, 10), (2, 70, 15)])
def test_taco2_fp32_forward_backward_matches_oracle(dev, shape):
    N, Ti, To = shape
    hp = small_hparams()
    m = _model(hp, "fp32")
    params, stats = m.numpy_params(), m.numpy_stats()
    # inputs on which the fp32 forward pass and the float64 oracle take the same branch at every ReLU (and no L1 term
    # sits within noise of its sign change): EVERY gradient can then be held to the max-norm bound.  The BatchNorm sums
    # are deterministic (ns_gemm stat_part), so the choice of the batch is repeatable.
    inputs, lengths, mel, lin = same_branch_batch(m, hp, N, Ti, To, seed=N)
    out, (loss, mel_loss, lin_loss), grads = oracle_run(hp, params, stats, inputs, lengths, mel, lin)
    m.initialize(inputs, lengths, None, mel, lin)
    m.backward()
    m.read_losses()
    torch.cuda.synchronize()
    ftol = 5e-4
    assert _rel(m.decoder_outputs.cpu().numpy(), out["decoder_outputs"].detach().numpy()) < 2e-4
    assert _rel(m.alignments.cpu().numpy(), out["alignments"].detach().numpy()) < 2e-4
    assert _rel(m.mel_outputs.cpu().numpy(), out["mel_outputs"].detach().numpy()) < ftol
    assert _rel(m.linear_outputs.cpu().numpy(), out["linear_outputs"].detach().numpy()) < ftol
    ltol = 1e-5
    assert abs(m.loss - loss) < ltol * max(1.0, abs(loss))
    assert abs(m.mel_loss - mel_loss) < ltol and abs(m.linear_loss - lin_loss) < ltol
    got = m.numpy_grads()
    bad = []
    for k in grads:
        scale = np.abs(grads[k]).max()
        err = np.abs(got[k] - grads[k]).max()
        # fp32 on the GPU vs float64 on the CPU through ~10 BatchNorms over a few hundred samples; typical error is
        # 1e-4 of the tensor's scale, the bound leaves room for the worst tensor (a ReLU on the wrong side of its kink
        # shows as 1e-2 .. 3e-1)
        if err > 2e-3 * scale + 5e-6:      # the floor covers conv biases in front of BatchNorm (true gradient 0)
            bad.append((k, float(err), float(scale)))
    print("gradient tensors over the bound:", bad)
    assert not bad, bad
    # BatchNorm moving statistics (UPDATE_OPS)
    st = m.numpy_stats()
    for k, v in out["bn_updates"].items():
        assert np.abs(st[k] - v.numpy()).max() < 1e-4, k


def _grad_summary(rep):
    worst = sorted(rep["grad"].items(), key=lambda kv: -kv[1][0])
    med = float(np.median([v[0] for v in rep["grad"].values()]))
    return worst[0][0], worst[0][1][0], med


def test_taco2_bf16_within_north_star_tolerance(dev):
    N, Ti, To = 4, 16, 40
    hp = small_hparams()
    m = _model(hp, "bf16")
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=7)
    mel, lin = stabilise_targets(hp, m.numpy_params(), m.numpy_stats(), inputs, lengths, mel, lin)
    rep = oracle_report(m, hp, inputs, lengths, mel, lin)
    print("bf16 small: ReLU branch differences", check_flips(rep, "bf16"))
    # single-pass bf16 operands: ~1e-2 relative per O(1) output (measured 1.4e-2 here); the
    # north_star 1e-3 tolerance is met by the split-bf16 (3-pass) mode, tested separately
    assert rep["out"]["mel_outputs"][2] < 3e-2, rep["out"]["mel_outputs"]
    assert abs(rep["loss"][0] - rep["loss"][1]) < 5e-3 * abs(rep["loss"][1])
    # every gradient tensor in relative L2 against the oracle on the same ReLU branches (round 2 held them to a cosine
    # > 0.9 only); measured at this shape: worst tensor 0.17 (the attention query layer), median 0.083
    name, worst, med = _grad_summary(rep)
    print("bf16 small: worst gradient tensor %s rel L2 %.3e, median %.3e" % (name, worst, med))
    assert worst < 0.35 and med < 0.15, (name, worst, med)


def test_taco2_adam_step_matches_oracle(dev):
    from oracle import taco2_oracle as O
    N, Ti, To = 2, 9, 15
    hp = small_hparams()
    m = _model(hp, "fp32")
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=11)
    params, stats = m.numpy_params(), m.numpy_stats()
    mel, lin = stabilise_targets(hp, params, stats, inputs, lengths, mel, lin)
    _, _, grads = oracle_run(hp, params, stats, inputs, lengths, mel, lin)
    m.add_optimizer(global_step=0)
    m.step(inputs, lengths, mel, lin)
    g = {k: torch.tensor(v, dtype=torch.float64) for k, v in grads.items()}
    g, gn = O.clip_by_global_norm(g, 1.0)
    p = {k: torch.tensor(v, dtype=torch.float64) for k, v in params.items()}
    mm = {k: torch.zeros_like(v) for k, v in p.items()}
    vv = {k: torch.zeros_like(v) for k, v in p.items()}
    O.adam_step(p, g, mm, vv, 1, O.learning_rate(hp.values(), 0), hp.adam["beta1"], hp.adam["beta2"])
    new = m.numpy_params()
    assert abs(m.grad_norm - gn) < 1e-3 * gn
    for k in p:
        upd_ref = p[k].numpy() - params[k]
        upd_got = new[k] - params[k]
        # Adam's first step is lr*sign(g) for |g| >> eps; compare where the oracle gradient is not tiny
        mask = np.abs(g[k].numpy()) > 1e-6
        if mask.any():
            assert np.abs(upd_got[mask] - upd_ref[mask]).max() < 2e-4, k
    assert m.global_step == 1


@pytest.mark.parametrize("mode", ["bf16x3", "mixed"])
def test_taco2_split_bf16_meets_north_star_tolerance(dev, mode):
    """north_star: mel outputs within 1e-3 (mean L1) of the reference arithmetic, on bf16 MFMA.
    Products on the mel path run as three split-bf16 MFMA passes (x = hi + lo)."""
    N, Ti, To = 4, 16, 40
    hp = small_hparams()
    m = _model(hp, mode)
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=7)
    mel, lin = stabilise_targets(hp, m.numpy_params(), m.numpy_stats(), inputs, lengths, mel, lin)
    rep = oracle_report(m, hp, inputs, lengths, mel, lin)
    print("%s small: ReLU branch differences" % mode, check_flips(rep, mode))
    assert rep["out"]["mel_outputs"][2] < 1e-3, rep["out"]["mel_outputs"]
    assert rep["out"]["alignments"][1] < 1e-3, rep["out"]["alignments"]
    assert abs(rep["mel_loss"][0] - rep["mel_loss"][1]) < 2e-3 * abs(rep["mel_loss"][1])
    # in 'mixed' the mel->linear expand net runs single-pass bf16, so the linear loss moves by ~1e-2
    assert abs(rep["loss"][0] - rep["loss"][1]) < (2e-3 if mode == "bf16x3" else 3e-2) * abs(rep["loss"][1])
    # every gradient tensor in relative L2 on the oracle's side of the GPU's own ReLU branches (round 2: cosine > 0.999 /
    # 0.8).  'mixed' back-propagates with single-pass bf16 products and runs the expand net in bf16.
    name, worst, med = _grad_summary(rep)
    print("%s small: worst gradient tensor %s rel L2 %.3e, median %.3e" % (mode, name, worst, med))
    wb, mb = (6e-4, 2e-4) if mode == "bf16x3" else (0.15, 4e-2)     # measured 1.7e-4 / 6.9e-5 and 6.7e-2 / 1.7e-2
    assert worst < wb and med < mb, (name, worst, med)


def test_taco2_mixed_full_width_forward(dev):
    """The shipped layer widths (512-channel convs, 1024-unit decoder LSTMs, 1025 bins) at a short
    length: mel L1 < 1e-3 against the float64 oracle in the benchmarked precision mode."""
    from nspeech_amd import hparams as hparams_mod
    hp = hparams_mod.load("taco2")
    N, Ti, To = 2, 24, 40
    m = _model(hp, "mixed", seed=5)
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=9)
    out, _, _ = oracle_run(hp, m.numpy_params(), m.numpy_stats(), inputs, lengths, mel, lin, need_grad=False)
    m.initialize(inputs, lengths, None, mel, lin)
    l1 = np.abs(m.mel_outputs.float().cpu().numpy() - out["mel_outputs"].detach().numpy()).mean()
    assert l1 < 1e-3, l1
    l1d = np.abs(m.decoder_outputs.float().cpu().numpy() - out["decoder_outputs"].detach().numpy()).mean()
    assert l1d < 1e-3, l1d


def test_taco2_full_size_properties(dev):
    """BASELINE config C2 (batch 32, T_in 160, T_out 1000, shipped widths) meets the float64 oracle in
    tests/test_taco2_fullwidth_gpu.py (the benchmarked `mixed` launch, forward and backward, about two minutes of host
    time); here the same shape goes through size-independent properties that need no oracle pass:
    (1) the benchmarked `mixed` mode stays within north_star's 1e-3 mel L1 of the exact-fp32 GPU mode (which the
        small-size tests pin to the oracle), and the two losses agree;
    (2) a second forward pass reproduces the first up to the order of the fp32 atomic sums (BatchNorm statistics);
    (3) the gradient is the derivative of the loss: a central difference along the gradient direction matches
        |g|^2-scaled prediction (split-bf16 x3 mode, whose forward and backward are both ~fp32-accurate)."""
    from nspeech_amd import hparams as hparams_mod
    hp = hparams_mod.load("taco2")
    N, Ti, To = 32, 160, 1000
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=11)
    ref = _model(hp, "fp32", seed=5)
    ref.initialize(inputs, lengths, None, mel, lin)
    ref.backward()
    ref.read_losses()
    ref_mel = ref.mel_outputs.float().clone()
    ref_loss, ref_mel_loss = ref.loss, ref.mel_loss
    del ref
    torch.cuda.empty_cache()

    m = _model(hp, "mixed", seed=5)
    m.initialize(inputs, lengths, None, mel, lin)
    first = m.mel_outputs.float().clone()
    l1 = (first - ref_mel).abs().mean().item()
    assert l1 < 1e-3, l1
    m.backward()
    m.read_losses()
    assert abs(m.mel_loss - ref_mel_loss) < 2e-3 * abs(ref_mel_loss)
    assert abs(m.loss - ref_loss) < 3e-2 * abs(ref_loss)
    m.initialize(inputs, lengths, None, mel, lin)
    again = (m.mel_outputs.float() - first).abs()
    assert again.mean().item() < 1e-4 and again.max().item() < 2e-3, (again.mean().item(), again.max().item())
    del m
    torch.cuda.empty_cache()

    x = _model(hp, "bf16x3", seed=5)
    x.initialize(inputs, lengths, None, mel, lin)
    x.backward()
    x.read_losses()
    g = x.flat_g.clone()
    gn = float(g.norm().item())
    p0 = x.flat_p.clone()
    eps = 2e-3 * x.loss / (gn * gn)            # predicted change of the loss: +-2e-3 * loss
    vals = []
    for sgn in (1.0, -1.0):
        x.flat_p.copy_(p0 + sgn * eps * g)
        x.refresh_shadows(full=True)
        x.initialize(inputs, lengths, None, mel, lin)
        x.backward()
        x.read_losses()
        vals.append(x.loss)
    slope = (vals[0] - vals[1]) / (2 * eps * gn * gn)
    assert 0.9 < slope < 1.1, (slope, vals)


def test_taco2_one_decoder_step_two_symbols(dev):
    """Smallest legal problem: one utterance, a two-symbol text, T_out = r (ONE decoder step).  BatchNorm then
    normalises over 5 frames, so only the parts in front of the first BatchNorm of the decoder side are held to the
    usual bound; everything must be finite and the step must run."""
    hp = small_hparams()
    m = _model(hp, "fp32")
    inputs, lengths, mel, lin = make_batch(hp, 1, 2, 5, seed=1)
    out, (loss, _, _), grads = oracle_run(hp, m.numpy_params(), m.numpy_stats(), inputs, lengths, mel, lin)
    m.add_optimizer(0)
    m.initialize(inputs, lengths, None, mel, lin)
    m.backward()
    m.read_losses()
    assert tuple(m.alignments.shape) == (1, 2, 1)
    assert _rel(m.decoder_outputs.cpu().numpy(), out["decoder_outputs"].detach().numpy()) < 2e-4
    assert _rel(m.alignments.cpu().numpy(), out["alignments"].detach().numpy()) < 2e-4
    assert _rel(m.mel_outputs.cpu().numpy(), out["mel_outputs"].detach().numpy()) < 5e-3
    assert abs(m.loss - loss) < 1e-3 * abs(loss)
    assert all(np.isfinite(v).all() for v in m.numpy_grads().values())
    m.apply_gradients()
    assert np.isfinite(m.flat_p.cpu().numpy()).all()


def test_taco2_full_size_mixed_gradients_follow_split_bf16(dev):
    """BASELINE config C2 shapes: the gradients of the benchmarked `mixed` mode (single-pass bf16 backward on bf16
    copies of the gradients / layer inputs, 256-tile data-gradient kernel, batched attention products) point the same
    way as those of the split-bf16 x3 mode, tensor by tensor, and the global norms agree."""
    from nspeech_amd import hparams as hparams_mod
    hp = hparams_mod.load("taco2")
    N, Ti, To = 32, 160, 1000
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=13)
    grads = {}
    for mode in ("bf16x3", "mixed"):
        m = _model(hp, mode, seed=5)
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
        m.read_losses()
        grads[mode] = (m.numpy_grads(), m.loss)
        del m
        torch.cuda.empty_cache()
    (ga, la), (gb, lb) = grads["bf16x3"], grads["mixed"]
    assert abs(la - lb) < 3e-2 * abs(la)
    na = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in ga.values()))
    nb = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in gb.values()))
    assert abs(na - nb) < 0.1 * na, (na, nb)
    # per tensor, relative L2 (round 2: cosine > 0.7).  The two GPU passes take their own ReLU / L1-sign branches (the
    # expand net runs in bf16 in `mixed`), so unlike the oracle tests at short lengths this bound includes kink flips.
    rel = {}
    for k in ga:
        a, b = ga[k].ravel().astype(np.float64), gb[k].ravel().astype(np.float64)
        if np.linalg.norm(a) < 1e-7 * na or k.endswith("conv1d/bias"):
            continue        # a bias in front of BatchNorm has a (near-)zero true gradient
        rel[k] = float(np.linalg.norm(a - b) / np.linalg.norm(a))
    worst = sorted(rel.items(), key=lambda kv: -kv[1])[:4]
    med = float(np.median(list(rel.values())))
    print("full size, mixed vs bf16x3: worst tensors %s, median %.3e, |g| %.4e vs %.4e" % (worst, med, nb, na))
    assert worst[0][1] < 0.1 and med < 1.5e-2, (worst, med)        # measured 3.8e-2 (an encoder BatchNorm beta) / 5.0e-3


@pytest.mark.parametrize("mode", ["fp32", "mixed"])
def test_taco2_two_passes_are_bitwise_repeatable(dev, mode):
    """BatchNorm batch statistics and the BatchNorm-backward sums are added up in a fixed order (no float atomics), so
    two passes over the same inputs give the same bits for every output, for the gradient that flows through the
    network and - since the split-K products, bias sums and table rows add in a fixed order too - for every weight
    gradient."""
    N, Ti, To = 6, 30, 60
    hp = small_hparams()
    m = _model(hp, mode)
    m.deterministic = True          # hparams.deterministic_gradients: the split-K products park their partial tiles
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=21)
    runs = []
    for _ in range(2):
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
        torch.cuda.synchronize()
        runs.append(dict(mel=m.mel_outputs.clone(), lin=m.linear_outputs.clone(), al=m.alignments.clone(),
                         dmel=m._bufs["d_mel"].clone(), dvalues=m._bufs["d_values"].clone(), g=m.flat_g.clone()))
    a, b = runs
    for k in ("mel", "lin", "al", "dmel", "dvalues"):
        assert torch.equal(a[k], b[k]), k
    # and for the whole gradient (round 4): split-K products, bias sums, the embedding table's rows and the attention
    # post-pass add in a fixed order - no float atomic is left between the loss and flat_g
    g0, g1 = a["g"], b["g"]
    for name, (off, shape) in m.layout.entries.items():
        n = int(np.prod(shape))
        assert torch.equal(g0[off:off + n], g1[off:off + n]), (name, (g0[off:off + n] - g1[off:off + n]).abs().max().item())


def test_taco2_default_atomic_split_k_agrees_to_rounding(dev):
    """The default (hparams.deterministic_gradients false): split-K products meet in fp32 atomics, so two passes agree in
    everything but the last bits of those sums; the deterministic pass lies within the same rounding of both."""
    N, Ti, To = 6, 30, 60
    hp = small_hparams()
    m = _model(hp, "mixed")
    assert m.deterministic is False
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=21)
    gs = []
    for det in (False, False, True):
        m.deterministic = det
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
        torch.cuda.synchronize()
        gs.append(m.flat_g.double().clone())
    for g1 in gs[1:]:
        for name, (off, shape) in m.layout.entries.items():
            if name.endswith("conv1d/bias"):
                continue
            n = int(np.prod(shape))
            d = (gs[0][off:off + n] - g1[off:off + n]).abs().max().item()
            sc = gs[0][off:off + n].abs().max().item()
            assert d <= 2e-5 * sc + 1e-9, (name, d, sc)


@pytest.mark.parametrize("size", ["small", "full"])
def test_taco2_weight_gradients_on_the_second_stream_change_nothing(dev, size):
    """The decoder / attention / postnet weight gradients that run on a second stream beside the encoder BiLSTM
    (Tacotron2.overlap_wgrads) equal the ones launched in line BIT FOR BIT (the split-K sums have a fixed order since
    round 4) - at the benchmark shape too, where a buffer reused too early would show."""
    from nspeech_amd import hparams as hparams_mod
    if size == "small":
        hp, (N, Ti, To) = small_hparams(), (6, 30, 60)
    else:
        hp, (N, Ti, To) = hparams_mod.load("taco2"), (32, 160, 1000)
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=17)
    m = _model(hp, "mixed", seed=7)
    m.deterministic = True
    gs = []
    for overlap in (False, True, True):
        m.overlap_wgrads = overlap
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
        torch.cuda.synchronize()
        assert not m._deferred
        gs.append(m.flat_g.clone())
    assert m._side is not None           # the second stream was really used
    for g1 in gs[1:]:
        for name, (off, shape) in m.layout.entries.items():
            n = int(np.prod(shape))
            assert torch.equal(gs[0][off:off + n], g1[off:off + n]), (name, (gs[0][off:off + n] - g1[off:off + n]).abs().max().item())


@pytest.mark.parametrize("mode", ["bf16", "mixed"])
def test_taco2_full_width_backward_repeats_over_many_launches(dev, mode):
    """The persistent kernels start their workgroups at slightly different times from launch to launch; nothing in the
    results may depend on that.  Thirty forward + backward passes at the shipped widths (the cluster kernels need
    them) must agree bit for bit in every buffer of the backward chain and in the whole gradient.  (Found this way: a missing barrier in front of the attention
    backward's first history fill gave one workgroup a zero alignment once in 20 - 50 launches.)"""
    from nspeech_amd import hparams as hparams_mod
    hp = hparams_mod.load("taco2")
    N, Ti, To = 8, 48, 100
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=3)
    m = _model(hp, mode, seed=5)
    m.deterministic = True
    m.overlap_wgrads = False            # one stream: every buffer has its final contents when the pass returns
    exact = ("d_energy", "d_q", "d_ga", "d_p2", "d_f1", "d_ctx_t", "d_hc", "d_h1", "d_h2", "d_keys_t", "d_enc_a",
             "d_enc_b", "d_act_a", "d_act_b", "d_mel")
    ref = None
    for run in range(30):
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
        torch.cuda.synchronize()
        m.check_status()
        snap = {k: m._bufs[k].clone() for k in exact if k in m._bufs}
        g = m.flat_g.clone()
        if ref is None:
            ref, gref = snap, g
            assert len(ref) >= 10
            continue
        for k, v in snap.items():
            assert torch.equal(v, ref[k]), (mode, run, k, (v.float() - ref[k].float()).abs().max().item())
        assert torch.equal(g, gref), (mode, run, (g - gref).abs().max().item())


def test_model_audio_is_griffin_lim_of_the_linear_outputs(dev):
    """tacotron.py:107 / train.py:100-102: model.audio[0] is the in-graph Griffin-Lim of linear_outputs[0]."""
    from nspeech_amd.utils import audio as A
    hp = small_hparams(frame_length_ms=5.0, frame_shift_ms=1.25)      # n_fft 128, window 100, hop 25 at num_freq 65
    m = _model(hp, "fp32")
    inputs, lengths, mel, lin = make_batch(hp, 2, 9, 20, seed=1)
    m.initialize(inputs, lengths, None, mel, lin)
    assert len(m.audio) == 2
    w0 = m.audio[0]
    want = A.griffin_lim_gpu(m.linear_outputs[0].float().contiguous())
    assert w0.shape == ((20 - 1) * 25 + 100,) and torch.equal(w0, want)
    assert torch.equal(m.audio.all()[1], m.audio[1])
