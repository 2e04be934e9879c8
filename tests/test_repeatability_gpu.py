"""Launch-to-launch repeatability of the paths outside the training step: the same call, repeated in one process,
must return the same bits.  None of these kernels adds with float atomics, so any difference is a race (a barrier
missing between waves, an exchange read too early).  The training step has its own test
(test_taco2_gpu.py::test_taco2_full_width_backward_repeats_over_many_launches)."""
import numpy as np
import pytest
import torch

from test_audio_oracle import _speechlike
from util import make_batch

pytestmark = pytest.mark.gpu


def test_griffin_lim_repeats(dev):
    from nspeech_amd import hparams
    from nspeech_amd.utils import audio as A
    hparams.load("taco2")
    y = _speechlike(200000, 7)
    spec = A.spectrogram(y).T[:797].copy()
    one = A.griffin_lim_gpu(spec).clone()                       # wave-per-frame kernel, 60 iterations
    batch = np.stack([spec] * 6)
    many = A.griffin_lim_gpu(batch).clone()
    for _ in range(8):
        assert torch.equal(A.griffin_lim_gpu(spec), one)
        assert torch.equal(A.griffin_lim_gpu(batch), many)
    assert torch.equal(many.view(6, -1)[3], one.view(-1))       # a clip does not depend on its neighbours in the batch


def test_synthesis_repeats(dev):
    """Free-running decoder at the shipped widths (persistent encoder kernels, per-step decoder, HIP-graph replay)."""
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    hp = hparams_mod.load("taco2")
    hp.max_iters = 40
    inputs, lengths, _, _ = make_batch(hp, 4, 60, 10, seed=9)
    for mode in ("mixed", "bf16"):
        m = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=4)
        ref = None
        for run in range(10):
            m.initialize(inputs, lengths)
            torch.cuda.synchronize()
            got = (m.mel_outputs.clone(), m.linear_outputs.clone(), m.alignments.clone())
            if ref is None:
                ref = got
            for a, b in zip(got, ref):
                assert torch.equal(a, b), (mode, run)


def test_wavenet_generation_repeats(dev):
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    from nspeech_amd.models.wavenet import receptive_field
    hp = hparams_mod.load("wavenet")
    rf = receptive_field(hp)
    m = create_model("simple_wavenet", hp, device="cuda:0", dtype="bf16", seed=6)
    seeds = np.random.default_rng(1).integers(0, hp.quantization_channels, (3, rf + 5)).astype(np.int32)
    un = np.random.default_rng(2).random((3, 48))
    ref = m.generate(seeds, 48, uniforms=un).clone()
    for _ in range(6):
        assert torch.equal(m.generate(seeds, 48, uniforms=un), ref)


def test_taco1_training_pass_repeats(dev):
    """Tacotron-1 at its shipped widths (CBHG banks, highway stack, BiGRU, GRU attention / decoder cells)."""
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd.models import create_model
    hp = hparams_mod.load("taco1")
    inputs, lengths, mel, lin = make_batch(hp, 4, 30, 40, seed=11)
    for mode in ("fp32", "bf16"):
        m = create_model("taco1", hp, device="cuda:0", dtype=mode, seed=3)
        ref = None
        for run in range(8):
            m.initialize(inputs, lengths, None, mel, lin)
            m.backward()
            torch.cuda.synchronize()
            got = (m.mel_outputs.clone(), m.linear_outputs.clone(), m.alignments.clone(), m.flat_g.clone())
            if ref is None:
                ref = got
            for a, b in zip(got[:3], ref[:3]):
                assert torch.equal(a, b), (mode, run)
            # weight gradients: split-K and scatter sums add with float atomics
            assert (got[3] - ref[3]).abs().max().item() <= 2e-5 * ref[3].abs().max().item(), (mode, run)
