import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The float64 oracle runs on torch-CPU.  A GPU box shows every core of its host but grants a 16-core share per GPU:
    # with one thread per visible core the oracle's many small ops crawl (the full-length parity test at N = 8 took 249 s
    # that way, 4x what 8 cores need) - keep to the share, as bench.py's cpu_baseline does.
    try:
        import torch
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(cores, 16)))
    except Exception:
        pass


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _reset_product_modes():
    """ops.F32_PASSES is process-wide state that a model sets for its own precision mode: a test that calls ops.gemm
    on fp32 operands directly must not inherit the mode of whatever model ran before it."""
    try:
        from nspeech_amd import ops
        ops.F32_PASSES = 0
    except Exception:
        pass
    yield
