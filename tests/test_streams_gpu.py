"""ns_streams_concurrent / ops.concurrent_stream: HIP deals its hardware queues to streams in turn, so every
queue-count-th stream of a process shares the current stream's queue and runs BEHIND it - an overlap planned on such a
stream (the deferred weight gradients of Tacotron-2, the pipelined GRUs of Tacotron-1) is silently lost.  The probe
finds out with two one-thread kernels; the models take their second streams through it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_probe_and_picked_stream(dev):
    from nspeech_amd import ops
    cur = torch.cuda.current_stream()
    assert not ops.streams_concurrent(cur, cur)                  # one stream is never beside itself
    keep = [torch.cuda.Stream() for _ in range(16)]
    pattern = [ops.streams_concurrent(cur, s) for s in keep]
    print("\nstreams 1..16 of this process beside the current one:", "".join("c" if c else "S" for c in pattern))
    if not any(pattern):                                         # one hardware queue (GPU_MAX_HW_QUEUES=1?): nothing to pick from
        pytest.skip("no second hardware queue on this device / runtime configuration")
    s = ops.concurrent_stream(torch.device("cuda:0"))
    assert ops.streams_concurrent(cur, s)
    # and the probe does what it says: a kernel queued on a NON-concurrent stream starts only after the current stream's
    shared = [k for k, c in zip(keep, pattern) if not c]
    if shared:                                                   # (every 8th stream on ROCm 7.2: usually one or two of 16)
        assert not ops.streams_concurrent(cur, shared[0])


def test_models_take_probed_side_streams(dev):
    from util import make_batch, small_hparams
    from nspeech_amd import ops
    from nspeech_amd.models import create_model
    burn = [torch.cuda.Stream() for _ in range(6)]               # the situation of a process that has made streams before
    x = torch.zeros(16, device="cuda:0")
    for s in burn:
        with torch.cuda.stream(s):
            ops.zero(x)
    torch.cuda.synchronize()
    hp = small_hparams()
    m = create_model("taco2", hp, device="cuda:0", dtype="bf16", seed=3)
    m.add_optimizer(0)
    inputs, lengths, mel, lin = make_batch(hp, 3, 12, 20, seed=11)
    m.step(inputs, lengths, mel, lin)
    if m._side is not None:
        assert ops.streams_concurrent(torch.cuda.current_stream(), m._side)
