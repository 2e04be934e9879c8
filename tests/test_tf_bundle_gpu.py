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
