"""The oracle against the committed fixtures (tests/golden/op_fixtures.npz, written by make_op_fixtures.py): an edit
to oracle/*.py that changes what it computes shows up here, on every CPU run, instead of moving the oracle and the
kernels together (VERDICT r3 missing #6, SURVEY 8c iii)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

FIX = np.load(os.path.join(ROOT, "tests", "golden", "op_fixtures.npz"))


def _close(a, b, tol=2e-6):          # the fixtures are stored as float32
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), np.abs(a - b).max()


def test_oracle_reproduces_every_fixture():
    import make_op_fixtures as M
    fresh = {}
    fresh.update(M.taco2_case())
    fresh.update(M.taco1_case())
    fresh.update(M.audio_case())
    assert set(fresh) == set(FIX.files), sorted(set(fresh) ^ set(FIX.files))
    for k in FIX.files:
        if k.endswith("grad_names"):
            assert list(fresh[k]) == list(FIX[k])
        elif k.endswith(("digest", "losses")):
            _close(fresh[k], FIX[k], 1e-10)          # kept in float64
        else:
            _close(fresh[k], FIX[k])


def test_fixture_sanity():
    """Known structure of the stored results: shapes of SURVEY 8c iii's case (N = 2, T_in = 11, T_out = 20), alignment
    rows are distributions, the saturated features of the shipped min_level_db against the live ones."""
    assert FIX["taco2/mel_outputs"].shape == (2, 20, 16) and FIX["taco2/linear_outputs"].shape == (2, 20, 65)
    al = FIX["taco2/alignments"]
    assert al.shape == (2, 11, 4) and np.abs(al.sum(axis=1) - 1.0).max() < 1e-6
    assert FIX["taco2_infer/mel_outputs"].shape == (2, 30, 16)
    assert abs(float(FIX["taco2/losses"][0]) - float(FIX["taco2/losses"][1]) - float(FIX["taco2/losses"][2])) < 1e-12
    assert FIX["audio/spectrogram"].shape == (1025, 25) and FIX["audio/melspectrogram"].shape == (80, 25)
    assert (FIX["audio/spectrogram"] > 0.999).mean() > 0.99           # min_level_db = +100 saturates
    live = FIX["audio/spectrogram_min_level_db_-100"]
    assert 0.05 < live.std() and live.min() >= 0.0 and live.max() <= 1.0
