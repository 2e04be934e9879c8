"""Tacotron-1 multi-speaker sites (modules.py:157-169, rnn_wrappers.py:28-30) on the CPU: the parameter layout the
model allocates and the oracle's semantics for the initial state of the bidirectional GRU."""
import numpy as np
import torch

from nspeech_amd import hparams as hparams_mod
from nspeech_amd.models import params as P


def _hp(n_spk):
    hp = hparams_mod.load("taco1")
    for k, v in dict(num_mels=16, num_freq=65, embedding_dim=32, encoder_prenet=[32, 128], encoder_cbhg_banks=3,
                     attention_dim=32, decoder_dim=32, post_cbhg_banks=2, post_cbhg_bank_sizes=[32], max_iters=4,
                     num_speakers=n_spk).items():
        setattr(hp, k, v)
    return hp


def test_layout_gains_the_speaker_variables_only_when_multi_speaker():
    one, _ = P.taco1_layout(_hp(1), 149)
    assert "speaker/speaker_embed" not in one.entries and "encoder_cbhg/dense/kernel" not in one.entries
    assert one.shape("encoder_cbhg/highway_3/highway/H/kernel") == (128, 128)
    assert one.shape("decoder/attention_gru/gates/kernel") == (128 + 32, 64)
    hp = _hp(4)
    tr, _ = P.taco1_layout(hp, 149)
    sd = hp.speaker_embed_dim
    assert tr.shape("speaker/speaker_embed") == (4, sd)
    for i, w in enumerate((256, 512, 1024, 2048)):
        assert tr.shape("encoder_cbhg/highway_%d/dense/kernel" % i) == (sd, w // 2)
        assert tr.shape("encoder_cbhg/highway_%d/highway/T/kernel" % i) == (w, w)
    assert tr.shape("encoder_cbhg/dense/kernel") == (sd, 128)
    assert tr.shape("encoder_cbhg/bidirectional_rnn/fw/gru_cell/candidate/kernel") == (2048 + 128, 128)
    assert tr.shape("decoder/dense/kernel") == (sd, 128)
    assert tr.shape("decoder/attention_gru/candidate/kernel") == (128 + 128 + 32, 32)
    assert "post_cbhg/highway_0/dense/kernel" not in tr.entries          # tacotron.py:93: no speaker in the post CBHG


def test_oracle_bigru_initial_state_and_gradients():
    from oracle import taco1_oracle as O
    torch.manual_seed(0)
    N, T, C, H = 2, 5, 6, 4
    p = {}
    for d in ("fw", "bw"):
        pre = "s/%s/gru_cell" % d
        p[pre + "/gates/kernel"] = torch.randn(C + H, 2 * H, dtype=torch.float64)
        p[pre + "/gates/bias"] = torch.ones(2 * H, dtype=torch.float64)
        p[pre + "/candidate/kernel"] = torch.randn(C + H, H, dtype=torch.float64)
        p[pre + "/candidate/bias"] = torch.zeros(H, dtype=torch.float64)
    x = torch.randn(N, T, C, dtype=torch.float64)
    h0 = torch.randn(N, H, dtype=torch.float64, requires_grad=True)
    lengths = torch.tensor([5, 3])
    y = O.bigru(x, lengths, p, "s", H, h0)
    # the backward cell of the short row meets h0 at its last valid step, its outputs past the length are zero
    pre = "s/bw/gru_cell"
    want = O.gru_cell(x[1:2, 2], h0[1:2], p[pre + "/gates/kernel"], p[pre + "/gates/bias"], p[pre + "/candidate/kernel"],
                      p[pre + "/candidate/bias"])
    assert torch.allclose(y[1, 2, H:], want[0])
    assert float(y[1, 3:].detach().abs().max()) == 0.0
    y.sum().backward()
    assert float(h0.grad.abs().min()) > 0.0

    hp = _hp(3)
    tr, st = P.taco1_layout(hp, 149)
    pp, ss = P.init_values(tr, st, seed=2)
    q = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in pp.items()}
    q.update({k: torch.tensor(v, dtype=torch.float64) for k, v in ss.items()})
    inp = torch.randint(1, 100, (3, 6))
    mel = torch.rand(3, 10, 16, dtype=torch.float64)
    lin = torch.rand(3, 10, 65, dtype=torch.float64)
    out = O.taco1_forward(q, hp.values(), inp, torch.tensor([6, 3, 4]), mel, lin, speaker_ids=torch.tensor([2, 0, 2]))
    O.taco1_loss(hp.values(), out, mel, lin)[0].backward()
    g = q["speaker/speaker_embed"].grad.numpy()
    assert np.abs(g[0]).max() > 0 and np.abs(g[2]).max() > 0 and np.abs(g[1]).max() == 0      # speaker 1 is not in the batch
    for k in ("encoder_cbhg/highway_2/dense/kernel", "encoder_cbhg/dense/bias", "decoder/dense/kernel"):
        assert float(q[k].grad.abs().max()) > 0, k
