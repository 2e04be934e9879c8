"""Helper for test_bench_ranks_gpu.py (run under torch.distributed.run, gloo, ranks sharing one GPU): the gradient a
rank holds after the bucketed asynchronous all-reduce inside Tacotron2.backward() must equal the sum of the
gradients the ranks compute on their own batches - i.e. no bucket is handed to the collective before its last
writer has been enqueued, and nothing is reduced twice."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from nspeech_amd import parallel
    from nspeech_amd.models import create_model
    from util import make_batch, small_hparams
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    hp = small_hparams()
    model = create_model("taco2", hp, device="cuda:0", dtype="fp32", seed=7, world_size=world)
    parallel.broadcast_parameters(model, 0)
    inputs, lengths, mel, lin = make_batch(hp, 3, 9, 20, seed=100 + rank)
    # keep every |target - prediction| far from zero: the L1 gradient is sign(), and one flip under the rounding noise
    # of the fp32 atomic sums moves a bucket's gradient by ~1e-2.  Targets are placed around the GPU's own predictions.
    model.initialize(inputs, lengths, None, mel, lin)
    dec, mo, lo_ = (t.float().cpu().numpy() for t in (model.decoder_outputs, model.mel_outputs, model.linear_outputs))
    import numpy as np
    sgn = np.where((np.arange(mel.size).reshape(mel.shape) % 2) == 0, 1.0, -1.0).astype(np.float32)
    mel = (np.where(sgn > 0, np.maximum(dec, mo), np.minimum(dec, mo)) + 0.1 * sgn).astype(np.float32)
    sgl = np.where((np.arange(lin.size).reshape(lin.shape) % 2) == 0, 1.0, -1.0).astype(np.float32)
    lin = (lo_ + 0.1 * sgl).astype(np.float32)
    model.add_optimizer(0)
    # local gradient, no reducer
    model.initialize(inputs, lengths, None, mel, lin)
    model.backward()
    torch.cuda.synchronize()
    local = model.flat_g.clone()
    parts = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(parts, local)
    want = sum(parts)
    # the same step with the bucket hooks
    model.reducer = parallel.make_reducer(model)
    model.initialize(inputs, lengths, None, mel, lin)
    model.backward()
    model.reducer.wait()
    torch.cuda.synchronize()
    got = model.flat_g
    # The BatchNorm sums are added in a fixed order, so a second pass over the same inputs reproduces the gradient up to
    # the rounding of the final weight-gradient sums (split-K atomics): compare per PARAMETER TENSOR - a bias or a
    # BatchNorm gamma handed to the collective before its last writer ran, or reduced twice, is off by O(1) there
    # even when it is a negligible share of its bucket.
    err = 0.0
    for name, (off, shape) in model.layout.entries.items():
        n = 1
        for d in shape:
            n *= d
        w = want[off:off + n]
        e = (got[off:off + n] - w).abs().max().item()
        sc = w.abs().max().item()
        assert e <= 1e-4 * sc + 1e-8, (rank, name, e, sc)
        err = max(err, e / (sc + 1e-30))
    scale = 1.0
    # and the optimiser step leaves every rank with identical parameters
    model.apply_gradients()
    torch.cuda.synchronize()
    ps = [torch.zeros_like(model.flat_p) for _ in range(world)]
    dist.all_gather(ps, model.flat_p)
    assert all(torch.equal(ps[0], p) for p in ps[1:])
    if rank == 0:
        print("DP_CHECK_OK %g" % (err / (scale + 1e-30)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
