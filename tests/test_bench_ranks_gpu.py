"""bench.py's multi-rank flow (barriers, every rank running the instrumented steps, bucketed all-reduce inside the
backward pass, rank 0 printing ONE JSON line) rehearsed with two ranks sharing one GPU over gloo - the driver's
real runs use nccl = RCCL with one rank per GPU, which a one-GPU box cannot host."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_gpu(dev):
    """`python bench.py --gpus 2` on its own: bench.py starts the two ranks itself (as a child process, before any GPU
    call) and rank 0's line reports what took part."""
    env = dict(os.environ, NSPEECH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--t-out", "200", "--t-in", "40", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and len(res["rank_devices"]) == 2, res
    assert res["steps"] == 2 and res["config"]["global_batch"] == 16
    assert res["value"] > 0 and res["roofline"]["achieved"] > 0 and res["scaling"] == "weak"
    assert abs(res["value"] - 16 * 200 / (res["ms_per_step"] * 1e-3)) < 1e-6 * res["value"]


def test_bench_refuses_a_world_that_is_not_gpus(dev):
    """--gpus 2 inside a ONE-rank launch must fail loudly instead of printing n_gpus: 1 (VERDICT r3 missing #1)."""
    env = dict(os.environ, NSPEECH_DIST_BACKEND="gloo", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "WORLD_SIZE=1" in out.stderr, (out.returncode, out.stderr[-500:])
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bucketed_allreduce_inside_backward_matches_sum_of_rank_gradients(dev):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29654", os.path.join(ROOT, "tests", "dp_check.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        sys.stderr.write(out.stderr)          # full child tracebacks in the captured output
    assert out.returncode == 0 and "DP_CHECK_OK" in out.stdout, out.stdout[-1500:]


def test_nccl_backend_one_rank_full_step(dev):
    """RCCL itself (backend 'nccl'), one rank on the one GPU: bucket hand-off, RCCL's stream, wait(), check_status()."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_check.py")], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        sys.stderr.write(out.stderr)
    assert out.returncode == 0 and "NCCL_CHECK_OK" in out.stdout, out.stdout[-1500:]
