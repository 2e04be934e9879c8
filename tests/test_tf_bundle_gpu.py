"""TensorFlow-bundle export / import of a whole model: the exported bundle restores bit-identical
parameters and BatchNorm statistics into a fresh model, through Synthesizer.load as well, and synthesis agrees."""
import numpy as np
import pytest
import torch

from util import make_batch, small_hparams

pytestmark = pytest.mark.gpu


def test_export_import_round_trip_and_synthesis(dev, tmp_path):
    from nspeech_amd.models import create_model
    from nspeech_amd.synthesizer import Synthesizer
    from nspeech_amd.utils import tf_bundle as B
    hp = small_hparams(max_iters=5, num_speakers=2)
    a = create_model("taco2", hp, device="cuda:0", dtype="fp32", seed=21)
    inputs, lengths, mel, lin = make_batch(hp, 2, 7, 10, seed=4)
    a.add_optimizer(0)
    a.step(inputs, lengths, mel, lin, speaker_ids=np.array([0, 1]))      # moves the weights and the moving statistics
    prefix = str(tmp_path / "model.ckpt-1")
    B.export_model(a, prefix)
    b = create_model("taco2", hp, device="cuda:0", dtype="fp32", seed=99)
    rep = B.load_into_model(b, prefix)
    assert rep["missing"] == [] and rep["global_step"] == 1 and b.global_step == 1
    assert torch.equal(a.flat_p, b.flat_p) and torch.equal(a.flat_stats, b.flat_stats)
    spk = np.array([1, 0])
    a.use_graph = b.use_graph = False
    a.initialize(inputs, lengths, spk)
    b.initialize(inputs, lengths, spk)
    assert torch.equal(a.mel_outputs, b.mel_outputs)
    s = Synthesizer(hp, dtype="fp32").load(prefix, "taco2")              # the eval.py path takes the bundle prefix
    assert torch.equal(s.model.flat_p, a.flat_p)
    # a single-speaker model cannot take this bundle: shape mismatch is refused
    c = create_model("taco2", small_hparams(max_iters=5), device="cuda:0", dtype="fp32", seed=1)
    with pytest.raises(ValueError):
        B.load_into_model(c, prefix)


def test_training_checkpoint_with_adam_slots_resumes_the_same_trajectory(dev, tmp_path):
    """train.py:60,67-71: the reference saves and restores its optimizer slots with the model.  A bundle written with the
    Adam moments (`<variable>/Adam`, `/Adam_1`, beta powers) restores them: the next training step from the restored
    model equals the next step of the original bit for bit; without the slots the moments start afresh
    and it does not."""
    from nspeech_amd.models import create_model
    from nspeech_amd.utils import tf_bundle as B
    hp = small_hparams(max_iters=5)
    inputs, lengths, mel, lin = make_batch(hp, 2, 7, 10, seed=4)
    a = create_model("taco2", hp, device="cuda:0", dtype="fp32", seed=21)
    a.deterministic = True           # hparams.deterministic_gradients: "the same trajectory" is then meant bit for bit
    a.add_optimizer(0)
    for _ in range(3):
        a.step(inputs, lengths, mel, lin)
    full, bare = str(tmp_path / "model.ckpt-3"), str(tmp_path / "bare.ckpt-3")
    B.export_model(a, full, with_adam_slots=True)
    B.export_model(a, bare)
    a.step(inputs, lengths, mel, lin)
    outs = []
    for prefix in (full, bare):
        b = create_model("taco2", hp, device="cuda:0", dtype="fp32", seed=99)
        b.deterministic = True
        rep = B.load_into_model(b, prefix)
        assert (rep["adam_slots"] is not None) == (prefix == full) and b.global_step == 3
        b.add_optimizer(b.global_step)
        b.step(inputs, lengths, mel, lin)
        outs.append(b.flat_p.clone())
    # no float atomic is left between the loss and the update (round 4), so the restored model's next step IS the
    # original's, bit for bit; the one without the moments is off by the size of an update (~1e-3)
    d_full = (outs[0] - a.flat_p).abs().max().item()
    d_bare = (outs[1] - a.flat_p).abs().max().item()
    assert torch.equal(outs[0], a.flat_p), d_full
    assert d_bare > 1e-4, d_bare
