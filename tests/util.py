"""Shared helpers for the parity tests: small Tacotron-2 configurations, synthetic batches and
the oracle driver (torch-CPU float64 autograd over oracle/taco2_oracle.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from nspeech_amd import hparams as hparams_mod  # noqa: E402


def small_hparams(**over):
    hp = hparams_mod.load("taco2")
    small = dict(num_mels=16, num_freq=65, embedding_dim=32, encoder_conv_channels=64, encoder_lstm_units=32,
                 attention_dim=64, decoder_lstm_units=64, postnet_conv_channels=64, expand_conv_channels=64,
                 expand_lstm_units=32, max_iters=50)
    small.update(over)
    for k, v in small.items():
        setattr(hp, k, v)
    return hp


def make_batch(hp, N, Ti, To, seed=0, vocab=149):
    rng = np.random.RandomState(seed)
    lengths = rng.randint(max(2, Ti // 2), Ti + 1, size=N).astype(np.int32)
    lengths[0] = Ti
    inputs = np.zeros((N, Ti), np.int32)
    for n in range(N):
        inputs[n, :lengths[n] - 1] = rng.randint(2, 64, size=lengths[n] - 1)
        inputs[n, lengths[n] - 1] = 1
    mel = rng.uniform(0, 1, size=(N, To, hp.num_mels)).astype(np.float32)
    lin = rng.uniform(0, 1, size=(N, To, hp.num_freq)).astype(np.float32)
    return inputs, lengths, mel, lin


def oracle_run(hp, params, stats, inputs, lengths, mel, lin, dtype=torch.float64, need_grad=True, speaker_ids=None):
    """Forward + loss + gradients with the CPU oracle.  Returns (out dict, loss tuple, grads dict)."""
    sys.path.insert(0, os.path.join(ROOT))
    from oracle import taco2_oracle as O
    p = {k: torch.tensor(v, dtype=dtype, requires_grad=need_grad) for k, v in params.items()}
    p.update({k: torch.tensor(v, dtype=dtype) for k, v in stats.items()})
    hpd = hp.values()
    out = O.taco2_forward(p, hpd, torch.tensor(inputs), torch.tensor(lengths),
                          torch.tensor(mel, dtype=dtype), torch.tensor(lin, dtype=dtype),
                          speaker_ids=None if speaker_ids is None else torch.tensor(speaker_ids))
    loss, mel_loss, lin_loss = O.taco2_loss(hpd, out, torch.tensor(mel, dtype=dtype), torch.tensor(lin, dtype=dtype))
    grads = {}
    if need_grad:
        loss.backward()
        grads = {k: (p[k].grad.numpy() if p[k].grad is not None else np.zeros_like(params[k])) for k in params}
    return out, (float(loss.detach()), float(mel_loss.detach()), float(lin_loss.detach())), grads


def stabilise_targets(hp, params, stats, inputs, lengths, mel, lin, margin=2e-3, rounds=4, speaker_ids=None):
    """The L1 losses have a sign() gradient: an element whose prediction sits within rounding
    noise of its target flips sign between fp32-on-GPU and float64-on-CPU and perturbs every
    upstream gradient by a finite amount.  Move such targets away from the oracle's prediction
    (mel targets also feed the teacher-forced decoder, hence a few rounds)."""
    from oracle import taco2_oracle as O
    mel, lin = mel.copy(), lin.copy()
    p = {k: torch.tensor(v, dtype=torch.float64) for k, v in params.items()}
    p.update({k: torch.tensor(v, dtype=torch.float64) for k, v in stats.items()})
    for _ in range(rounds):
        with torch.no_grad():
            out = O.taco2_forward(p, hp.values(), torch.tensor(inputs), torch.tensor(lengths),
                                  torch.tensor(mel, dtype=torch.float64), torch.tensor(lin, dtype=torch.float64),
                                  speaker_ids=None if speaker_ids is None else torch.tensor(speaker_ids))
        dm = out["mel_outputs"].numpy() - mel
        dl = out["linear_outputs"].numpy() - lin
        bm, bl = np.abs(dm) < margin, np.abs(dl) < margin
        if not bm.any() and not bl.any():
            break
        mel[bm] -= 10 * margin
        lin[bl] -= 10 * margin
    return mel.astype(np.float32), lin.astype(np.float32)


def kink_margin(hp, params, stats, inputs, lengths, mel, lin, speaker_ids=None):
    """Smallest non-zero |ReLU pre-activation| of the oracle's forward pass on these inputs."""
    from oracle import taco2_oracle as O
    p = {k: torch.tensor(v, dtype=torch.float64) for k, v in params.items()}
    p.update({k: torch.tensor(v, dtype=torch.float64) for k, v in stats.items()})
    O.KINK_LOG = []
    try:
        with torch.no_grad():
            O.taco2_forward(p, hp.values(), torch.tensor(inputs), torch.tensor(lengths),
                            torch.tensor(mel, dtype=torch.float64), torch.tensor(lin, dtype=torch.float64),
                            speaker_ids=None if speaker_ids is None else torch.tensor(speaker_ids))
        return min(O.KINK_LOG)
    finally:
        O.KINK_LOG = None


def well_posed_batch(hp, params, stats, N, Ti, To, seed, margin=2e-5, tries=12):
    """A synthetic batch (targets stabilised) none of whose ReLU pre-activations lies within `margin` of the kink, so
    that a float32 run and the float64 oracle take the same side everywhere and the gradients can be compared in the
    max norm.  Walks the data seeds from `seed`; deterministic."""
    for s in range(seed, seed + tries):
        inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=s)
        mel, lin = stabilise_targets(hp, params, stats, inputs, lengths, mel, lin)
        if kink_margin(hp, params, stats, inputs, lengths, mel, lin) > margin:
            return inputs, lengths, mel, lin
    raise AssertionError("no well-posed batch within %d seeds" % tries)
