"""simple_wavenet (BASELINE config 4) against the float64 oracle: logits, loss, every gradient, one Adam
step, predict_proba, mu-law round trip; exact-fp32 mode for parity, bf16 mode within its rounding."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hp(**over):
    from nspeech_amd import hparams as hparams_mod
    hp = hparams_mod.load("wavenet")
    small = dict(dilations_depth=2, dilations_length=3, residual_channels=16, dilation_channels=16, skip_channels=32,
                 quantization_channels=64)
    small.update(over)
    for k, v in small.items():
        setattr(hp, k, v)
    return hp


def _audio(N, T, seed):
    rng = np.random.RandomState(seed)
    t = np.arange(T) / 16000.0
    return np.stack([0.6 * np.sin(2 * np.pi * rng.uniform(100, 900) * t + rng.uniform(0, 6)) + 0.05 * rng.randn(T)
                     for _ in range(N)]).astype(np.float32)


def _oracle(hp, params, ids):
    from oracle import wavenet_oracle as O
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    loss, logits = O.loss(p, hp.values(), torch.tensor(ids))
    loss.backward()
    # the last layer's dense kernel feeds nothing (its residual output is unused): no gradient -> zeros
    return float(loss), logits.detach().numpy(), {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in p.items()}


@pytest.mark.parametrize("shape,over", [((2, 40), {}), ((1, 17), {}), ((3, 75), dict(quantization_channels=256, skip_channels=64)),
                                        ((2, 60), dict(dilations_depth=1, dilations_length=5, residual_channels=32,
                                                       dilation_channels=32))])
def test_wavenet_fp32_matches_oracle(dev, shape, over):
    from nspeech_amd.models import create_model
    from nspeech_amd.models.wavenet import mu_law_encode, receptive_field
    hp = _hp(**over)
    N, T = shape
    assert T > receptive_field(hp)
    m = create_model("simple_wavenet", hp, device="cuda:0", dtype="fp32", seed=4)
    audio = _audio(N, T, seed=N)
    ids = mu_law_encode(audio, hp.quantization_channels)
    loss, logits, grads = _oracle(hp, m.numpy_params(), ids)
    m.initialize(audio)
    m.backward()
    got_loss = m.read_losses()
    assert np.abs(m.raw_output.cpu().numpy() - logits).max() < 2e-5 * max(1.0, np.abs(logits).max())
    assert abs(got_loss - loss) < 1e-5 * max(1.0, abs(loss))
    got = m.numpy_grads()
    for k in grads:
        scale = np.abs(grads[k]).max()
        assert np.abs(got[k] - grads[k]).max() < 1e-4 * scale + 1e-8, (k, np.abs(got[k] - grads[k]).max(), scale)


def test_wavenet_adam_step_and_predict_proba(dev):
    from nspeech_amd.models import create_model
    from nspeech_amd.models.wavenet import mu_law_encode
    from oracle import wavenet_oracle as O
    hp = _hp()
    hp.decay_learning_rate = True
    m = create_model("simple_wavenet", hp, device="cuda:0", dtype="fp32", seed=1)
    m.add_loss().add_optimizer(0).add_stats()
    audio = _audio(2, 50, seed=9)
    ids = mu_law_encode(audio, hp.quantization_channels)
    p0 = m.numpy_params()
    _, _, grads = _oracle(hp, p0, ids)
    loss = m.step(audio)
    assert np.isfinite(loss) and m.global_step == 1
    # first Adam step with bias correction = lr * g / (|g| + eps'), lr from the Noam schedule at step 0, after the clip
    gn = np.sqrt(sum((g ** 2).sum() for g in grads.values()))
    lr = hp.initial_learning_rate * 4000.0 ** 0.5 * min(1 * 4000.0 ** -1.5, 1.0)
    new = m.numpy_params()
    for k, g in grads.items():
        gc = g * min(1.0, 1.0 / gn)
        want = -lr * gc / (np.abs(gc) + 1e-8)
        mask = np.abs(gc) > 1e-5
        if mask.any():
            # lr at step 0 of the Noam schedule is 5e-7: the update itself is only ~60 fp32 ulps of a 0.1-sized weight
            assert np.abs((new[k] - p0[k])[mask] - want[mask]).max() < 6e-2 * lr, k
    # predict_proba on a waveform longer than the receptive field
    wav = ids[0, :30]
    pr = m.predict_proba(wav).cpu().numpy()
    ref = O.predict_proba({k: torch.tensor(v, dtype=torch.float64) for k, v in new.items()}, hp.values(), torch.tensor(wav)).numpy()
    assert abs(pr.sum() - 1.0) < 1e-5 and np.abs(pr - ref).max() < 1e-5


def test_wavenet_bf16_close_to_oracle_and_mu_law(dev):
    from nspeech_amd.models import create_model
    from nspeech_amd.models.wavenet import mu_law_decode, mu_law_encode
    from oracle import wavenet_oracle as O
    hp = _hp(residual_channels=32, dilation_channels=32, skip_channels=64)
    m = create_model("simple_wavenet", hp, device="cuda:0", dtype="bf16", seed=2)
    audio = _audio(4, 120, seed=3)
    ids = mu_law_encode(audio, hp.quantization_channels)
    assert np.array_equal(ids, O.mu_law_encode(audio, hp.quantization_channels)) and ids.min() >= 0 and ids.max() < 64
    back = mu_law_decode(ids, hp.quantization_channels)
    assert np.abs(back - np.clip(audio, -1, 1)).max() < 0.08          # 64-level mu-law quantisation error
    loss, logits, grads = _oracle(hp, m.numpy_params(), ids)
    m.initialize(audio)
    m.backward()
    assert abs(m.read_losses() - loss) < 2e-2 * abs(loss)
    got = m.numpy_grads()
    for k in grads:
        a, b = got[k].ravel().astype(np.float64), grads[k].ravel()
        if np.linalg.norm(b) > 1e-8:
            assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)) > 0.98, k


def test_wavenet_incremental_generation_matches_sliding_window(dev):
    """The persistent sample-by-sample kernel (per-layer rings) against the oracle that re-runs the whole network on
    the last receptive-field samples for every new sample, with the same uniform numbers behind the draws."""
    from nspeech_amd.models import create_model
    from nspeech_amd.models.wavenet import mu_law_encode, receptive_field
    from oracle import wavenet_oracle as O
    hp = _hp()
    rf = receptive_field(hp)
    m = create_model("simple_wavenet", hp, device="cuda:0", dtype="fp32", seed=6)
    # sharpen the distributions so that draws are not all near-uniform: scale the last layer
    p = m.numpy_params()
    p["wavenet/postprocessing/postprocess2"] = p["wavenet/postprocessing/postprocess2"] * 12.0
    m.load_numpy_params(p)
    seeds = mu_law_encode(_audio(2, rf + 7, seed=5), hp.quantization_channels)
    un = np.random.default_rng(3).random((2, 12))
    got = m.generate(seeds, 12, uniforms=un).cpu().numpy()
    pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
    for b in range(2):
        ref = O.generate(pt, hp.values(), seeds[b], un[b])
        assert np.array_equal(got[b], ref), (b, got[b, -12:], ref[-12:])
    # the distribution behind the last draw = predict_proba on the history before it
    pr = m.last_probs.view(2, -1)[0].cpu().numpy()
    want = m.predict_proba(got[0, :-1]).cpu().numpy()
    assert np.abs(pr - want).max() < 1e-5
    assert got.min() >= 0 and got.max() < hp.quantization_channels


@pytest.mark.parametrize("over", [dict(residual_channels=16, dilation_channels=16, skip_channels=32),
                                  dict(residual_channels=32, dilation_channels=32, skip_channels=64, dilations_length=4)])
def test_wavenet_single_wave_chain_matches_reference_kernel(dev, over):
    """The fast generation kernel (one wavefront walks the residual chain, transposed bf16 weight shadows fetched three
    layers ahead) against the straightforward kernel on the same bf16 weights: same draws, same last distribution."""
    from nspeech_amd.models import create_model
    from nspeech_amd.models.wavenet import mu_law_encode, receptive_field
    hp = _hp(**over)
    rf = receptive_field(hp)
    m = create_model("simple_wavenet", hp, device="cuda:0", dtype="bf16", seed=8)
    p = m.numpy_params()
    p["wavenet/postprocessing/postprocess2"] = p["wavenet/postprocessing/postprocess2"] * 10.0
    m.load_numpy_params(p)
    seeds = mu_law_encode(_audio(3, rf + 5, seed=2), hp.quantization_channels)
    un = np.random.default_rng(1).random((3, 40))
    fast = m.generate(seeds, 40, uniforms=un, engine=1).cpu().numpy()
    pf = m.last_probs.clone()
    slow = m.generate(seeds, 40, uniforms=un, fast=False).cpu().numpy()      # reference kernel, same bf16 weights
    ps = m.last_probs
    # identical operands, different summation order: a draw can only differ where u falls within ~1e-6 of a CDF step
    assert (fast != slow).mean() < 0.02, (fast != slow).sum()
    if np.array_equal(fast, slow):
        assert (pf - ps).abs().max().item() < 1e-5


def test_wavenet_mfma_chain_close_to_valu_chain(dev):
    """The MFMA generation kernel rounds the layer inputs to bf16 (the VALU chains keep them fp32 against bf16 weights):
    same history -> next-sample distribution within bf16 rounding; over a run most draws coincide and all are valid."""
    from nspeech_amd.models import create_model
    from nspeech_amd.models.wavenet import mu_law_encode, receptive_field
    hp = _hp(residual_channels=32, dilation_channels=32, skip_channels=64, dilations_length=4)
    rf = receptive_field(hp)
    m = create_model("simple_wavenet", hp, device="cuda:0", dtype="bf16", seed=8)
    p = m.numpy_params()
    p["wavenet/postprocessing/postprocess2"] = p["wavenet/postprocessing/postprocess2"] * 10.0
    m.load_numpy_params(p)
    seeds = mu_law_encode(_audio(3, rf + 9, seed=4), hp.quantization_channels)
    un = np.random.default_rng(2).random((3, 1))
    a = m.generate(seeds, 1, uniforms=un, engine=2).cpu().numpy()
    pa = m.last_probs.clone()
    b = m.generate(seeds, 1, uniforms=un, engine=1).cpu().numpy()
    pb = m.last_probs
    assert (pa - pb).abs().max().item() < 3e-2 and abs(float(pa.sum()) - 3.0) < 1e-4
    un = np.random.default_rng(5).random((3, 48))
    a = m.generate(seeds, 48, uniforms=un, engine=2).cpu().numpy()
    b = m.generate(seeds, 48, uniforms=un, engine=1).cpu().numpy()
    assert a.min() >= 0 and a.max() < hp.quantization_channels and np.array_equal(a[:, :rf + 9], seeds)
    assert (a == b).mean() > 0.5          # histories part ways at the first differing draw


def _shipped():
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models.wavenet import receptive_field
    hp = hparams_mod.load("wavenet")          # wavenet.yaml as shipped: 5 x 10 layers, R = Dc = 32, S = 512, Q = 256
    assert receptive_field(hp) == 5117
    return hp, 5117


def test_wavenet_shipped_config_matches_oracle(dev):
    """The configuration bench.py times (50 layers, receptive field 5117): logits, loss and every gradient of one clip
    of receptive field + 16 samples against the float64 oracle, in exact-fp32 mode."""
    from nspeech_amd.models import create_model
    from nspeech_amd.models.wavenet import mu_law_encode
    hp, rf = _shipped()
    m = create_model("simple_wavenet", hp, device="cuda:0", dtype="fp32", seed=4)
    audio = _audio(1, rf + 16, seed=12)
    ids = mu_law_encode(audio, hp.quantization_channels)
    loss, logits, grads = _oracle(hp, m.numpy_params(), ids)
    assert logits.shape == (1, 16, 256)
    m.initialize(audio)
    m.backward()
    got_loss = m.read_losses()
    assert np.abs(m.raw_output.cpu().numpy() - logits).max() < 5e-5 * max(1.0, np.abs(logits).max())
    assert abs(got_loss - loss) < 1e-5 * max(1.0, abs(loss))
    got = m.numpy_grads()
    for k in grads:
        scale = np.abs(grads[k]).max()
        assert np.abs(got[k] - grads[k]).max() < 2e-4 * scale + 1e-8, (k, np.abs(got[k] - grads[k]).max(), scale)


def test_wavenet_mfma_generator_matches_oracle_distribution_at_shipped_config(dev):
    """The engine bench.py times (MFMA chain, bf16 weights and bf16-rounded layer inputs) against the float64 oracle on
    the same history: the next-sample distribution behind the first drawn sample.  Stated bf16 bound: 50 layers, each
    rounding its 32-wide input to 8 significant bits (2^-9 relative), accumulate ~sqrt(50) * 2^-9 ~ 1.4e-2 relative on
    the skip sums; with the last layer sharpened 10x that is a few 1e-2 on the largest logits, so total variation
    distance < 5e-2 and every probability within 15 % of the largest one."""
    from nspeech_amd.models import create_model
    from nspeech_amd.models.wavenet import mu_law_encode
    from oracle import wavenet_oracle as O
    hp, rf = _shipped()
    m = create_model("simple_wavenet", hp, device="cuda:0", dtype="bf16", seed=8)
    p = m.numpy_params()
    p["wavenet/postprocessing/postprocess2"] = p["wavenet/postprocessing/postprocess2"] * 10.0
    m.load_numpy_params(p)
    seeds = mu_law_encode(_audio(2, rf + 8, seed=4), hp.quantization_channels)
    un = np.random.default_rng(2).random((2, 1))
    out = m.generate(seeds, 1, uniforms=un, engine=2).cpu().numpy()
    pa = m.last_probs.view(2, -1).cpu().numpy().astype(np.float64)
    pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in m.numpy_params().items()}
    for b in range(2):
        ref = O.predict_proba(pt, hp.values(), torch.tensor(seeds[b])).numpy()
        assert ref.max() > 4.0 / 256, ref.max()            # a distribution with structure, not the uniform one
        tv = 0.5 * np.abs(pa[b] - ref).sum()
        worst = np.abs(pa[b] - ref).max() / ref.max()
        print("wavenet MFMA generator vs oracle: TV %.3e, worst |dp| / max p %.3e" % (tv, worst))
        assert abs(pa[b].sum() - 1.0) < 1e-4 and tv < 5e-2 and worst < 0.15, (b, tv, worst)
        # the draw is the inverse CDF of the kernel's own distribution at the given uniform number
        c = np.cumsum(pa[b])
        assert abs(int(out[b, -1]) - int(min(np.searchsorted(c, un[b, 0] * c[-1], side="right"), 255))) <= 1


def test_wavenet_helper_engine_equals_the_chain_engine(dev):
    """Engine 3 = the MFMA chain with the post-processing products (relu -> post1 -> relu -> post2) on four helper
    workgroups per waveform that keep their quarter of both kernels in registers; the same sums in another order: the
    next-sample distribution agrees with engine 2's to fp32 rounding, the draws coincide, the status word stays 0 - at
    batch 1, 3 and 32 (160 workgroups resident), and again on the same model (the exchange region is per call)."""
    from nspeech_amd.models import create_model
    from nspeech_amd.models.wavenet import mu_law_encode
    hp, rf = _shipped()
    m = create_model("simple_wavenet", hp, device="cuda:0", dtype="bf16", seed=8)
    p = m.numpy_params()
    p["wavenet/postprocessing/postprocess2"] = p["wavenet/postprocessing/postprocess2"] * 10.0
    m.load_numpy_params(p)
    if m._helper(torch.device("cuda:0")) is None:
        pytest.skip("no stream runs beside the current one on this runtime configuration: engine 2 is what generate() takes")
    for B, n_new in ((1, 40), (3, 24), (32, 6), (3, 24)):
        seeds = mu_law_encode(_audio(B, rf + 5, seed=4 + B), hp.quantization_channels)
        un = np.random.default_rng(2).random((B, n_new))
        a = m.generate(seeds, n_new, uniforms=un).cpu().numpy()             # the default engine at the shipped widths
        assert m.last_engine == 3 and int(m.last_status.item()) == 0
        pa = m.last_probs.clone()
        b = m.generate(seeds, n_new, uniforms=un, engine=2).cpu().numpy()
        pb = m.last_probs
        agree = (a == b).all(axis=1)
        assert agree.mean() >= 0.6, (B, agree)                # (a draw on a CDF edge may fall the other way: histories part there)
        rows = np.nonzero(agree)[0]
        d = (pa.view(B, -1)[rows] - pb.view(B, -1)[rows]).abs().max().item()
        assert d < 1e-5, (B, d)
