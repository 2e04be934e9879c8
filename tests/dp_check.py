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
    model.reducer = parallel.GradReducer(model.flat_g, parallel.bucket_ranges(model.layout))
    model.initialize(inputs, lengths, None, mel, lin)
    model.backward()
    model.reducer.wait()
    torch.cuda.synchronize()
    got = model.flat_g
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= 1e-3 * scale + 1e-9, (rank, err, scale)      # run-to-run noise of the fp32 atomic sums is ~1e-5
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
