#!/usr/bin/env python3
"""Writes tests/golden/op_fixtures.npz: seeded inputs and the float64 oracle's outputs for the hot path's blocks
(SURVEY 8c iii), so that the oracle cannot drift unnoticed - tests/test_fixtures_cpu.py re-runs the oracle against this
file on every CPU run, tests/test_fixtures_gpu.py holds the kernels to the same numbers.

    python3 tests/golden/make_op_fixtures.py          (only when a fixture is MEANT to change; commit the diff)

The fixtures are outputs of THIS build's oracle (oracle/*.py: a restatement of the reference's TensorFlow / librosa
arithmetic, "parity unpinned" - TF 1.7 and librosa cannot run here); they pin the oracle in time, they are not
outputs of the reference.  The text path's fixtures ARE reference outputs (make_text_golden.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

AUDIO_HP = dict(num_mels=80, num_freq=1025, sample_rate=20000, frame_length_ms=50, frame_shift_ms=12.5,
                preemphasis=0.97, min_level_db=100, ref_level_db=20, power=1.5, griffin_lim_iters=60)


def speechlike(L, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(L) / 20000.0
    f0 = rng.uniform(90, 250)
    y = sum(np.sin(2 * np.pi * f0 * (h + 1) * t) / (h + 1) for h in range(5))
    y *= 0.5 + 0.5 * np.sin(2 * np.pi * 4 * t)
    y += rng.normal(0, 0.01, L)
    return (0.8 * y / np.abs(y).max()).astype(np.float32)


def digest(a):
    """What is stored of a large tensor: its sum, its sum of squares and 16 evenly spaced elements."""
    a = np.asarray(a, np.float64).ravel()
    idx = np.linspace(0, a.size - 1, 16).astype(np.int64) if a.size else np.zeros(0, np.int64)
    return np.concatenate([[a.sum(), (a * a).sum()], a[idx]])


def taco2_case():
    from util import make_batch, oracle_run, small_hparams
    from nspeech_amd.models import params as P
    from nspeech_amd.utils.text.symbols import symbols
    hp = small_hparams()
    lay, st = P.taco2_layout(hp, len(symbols))
    pv, sv = P.init_values(lay, st, 11)
    inputs, lengths, mel, lin = make_batch(hp, 2, 11, 20, seed=3)
    out, losses, grads = oracle_run(hp, pv, sv, inputs, lengths, mel, lin)
    d = {"taco2/inputs": inputs, "taco2/lengths": lengths, "taco2/mel_targets": mel, "taco2/linear_targets": lin,
         "taco2/param_digest": np.stack([digest(pv[k]) for k in sorted(pv)]),
         "taco2/losses": np.asarray(losses, np.float64)}
    for k in ("mel_outputs", "linear_outputs", "decoder_outputs", "alignments"):
        d["taco2/" + k] = out[k].detach().numpy()
    d["taco2/grad_digest"] = np.stack([digest(grads[k]) for k in sorted(grads)])
    d["taco2/grad_names"] = np.asarray(sorted(grads))
    # free-running synthesis, 6 steps
    hp.max_iters = 6
    from oracle import taco2_oracle as O
    p = {k: torch.tensor(v, dtype=torch.float64) for k, v in list(pv.items()) + list(sv.items())}
    with torch.no_grad():
        inf = O.taco2_forward(p, hp.values(), torch.tensor(inputs), torch.tensor(lengths))
    for k in ("mel_outputs", "linear_outputs", "alignments"):
        d["taco2_infer/" + k] = inf[k].numpy()
    return d


def taco1_case():
    from util import make_batch
    from nspeech_amd import hparams as H
    from nspeech_amd.models import params as P
    from nspeech_amd.utils.text.symbols import symbols
    from oracle import taco1_oracle as O
    hp = H.load("taco1")
    for k, v in dict(num_mels=16, num_freq=65, embedding_dim=32, encoder_prenet=[32, 128], encoder_cbhg_banks=4,
                     attention_dim=64, decoder_dim=64, post_cbhg_banks=3, post_cbhg_bank_sizes=[64], max_iters=50,
                     batch_size=2).items():
        setattr(hp, k, v)
    lay, st = P.taco1_layout(hp, len(symbols))
    pv, sv = P.init_values(lay, st, 12)
    inputs, lengths, mel, lin = make_batch(hp, 2, 11, 20, seed=4)
    p = {k: torch.tensor(v, dtype=torch.float64) for k, v in list(pv.items()) + list(sv.items())}
    with torch.no_grad():
        out = O.taco1_forward(p, hp.values(), torch.tensor(inputs), torch.tensor(lengths),
                              torch.tensor(mel, dtype=torch.float64), torch.tensor(lin, dtype=torch.float64))
        loss = O.taco1_loss(hp.values(), out, torch.tensor(mel, dtype=torch.float64), torch.tensor(lin, dtype=torch.float64))
    d = {"taco1/inputs": inputs, "taco1/lengths": lengths, "taco1/mel_targets": mel, "taco1/linear_targets": lin,
         "taco1/param_digest": np.stack([digest(pv[k]) for k in sorted(pv)]),
         "taco1/losses": np.asarray([float(x) for x in loss], np.float64)}
    for k in ("mel_outputs", "linear_outputs", "alignments"):
        d["taco1/" + k] = out[k].numpy()
    return d


def audio_case():
    from oracle import audio_oracle as AO
    y = speechlike(6000, 5)
    n_fft, hop, win = AO.stft_parameters(AUDIO_HP)
    lin = AO.spectrogram(y, AUDIO_HP)
    live = AO.spectrogram(y, dict(AUDIO_HP, min_level_db=-100))
    d = {"audio/wav": y, "audio/preemphasis": AO.preemphasis(y, 0.97), "audio/inv_preemphasis": AO.inv_preemphasis(y, 0.97),
         "audio/spectrogram": lin, "audio/melspectrogram": AO.melspectrogram(y, AUDIO_HP),
         "audio/spectrogram_min_level_db_-100": live,
         "audio/melspectrogram_min_level_db_-100": AO.melspectrogram(y, dict(AUDIO_HP, min_level_db=-100)),
         "audio/griffin_lim_3_iters": AO.inv_spectrogram_tensorflow(live.T[:20].copy(), dict(AUDIO_HP, min_level_db=-100), iters=3)}
    return d


def main():
    torch.set_num_threads(4)
    d = {}
    d.update(taco2_case())
    d.update(taco1_case())
    d.update(audio_case())
    out = os.path.join(HERE, "op_fixtures.npz")
    np.savez_compressed(out, **{k: (np.asarray(v, np.float32) if np.asarray(v).dtype == np.float64 and not k.endswith(("digest", "losses"))
                                    else np.asarray(v)) for k, v in d.items()})
    print("wrote %s (%d arrays, %d bytes)" % (out, len(d), os.path.getsize(out)))


if __name__ == "__main__":
    main()
