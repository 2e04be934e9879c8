"""Helper for test_bench_ranks_gpu.py: the data-parallel hand-off on the REAL backend (nccl = RCCL) with the one GPU a
test box has.  A one-rank group makes every all-reduce the identity, so the reduced gradient must equal the local one,
while everything else is what an 8-GPU run executes per rank: RCCL is loaded and initialised, each bucket is handed to
an asynchronous all-reduce on RCCL's stream the moment its last writer is enqueued on the compute stream, wait()
joins them before clip + Adam, and (bf16 mode, full-width model) the persistent BiLSTM cluster kernels of the encoder
backward run while the decoder bucket is in RCCL's kernels - check_status() must stay clean."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from nspeech_amd import hparams as hparams_mod
    from nspeech_amd import parallel
    from nspeech_amd.models import create_model
    from util import make_batch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29671")
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    hp = hparams_mod.load("taco2")          # shipped widths: the cluster kernels need 64-unit multiples
    for mode in ("bf16", "mixed"):
        model = create_model("taco2", hp, device="cuda:0", dtype=mode, seed=5, world_size=1)
        parallel.broadcast_parameters(model, 0)
        inputs, lengths, mel, lin = make_batch(hp, 8, 48, 100, seed=3)
        model.add_optimizer(0)
        model.initialize(inputs, lengths, None, mel, lin)
        model.backward()
        torch.cuda.synchronize()
        local = model.flat_g.clone()
        model.reducer = parallel.make_reducer(model, force=True)
        assert model.reducer.active
        for _ in range(3):                   # several steps: buckets of step k+1 are released while nothing of step k is left
            model.initialize(inputs, lengths, None, mel, lin)
            model.backward()
            assert len(model.reducer.pending) == 4, len(model.reducer.pending)
            model.reducer.wait()
            torch.cuda.synchronize()
            model.check_status()
            got = model.flat_g
            for name, (off, shape) in model.layout.entries.items():
                n = 1
                for d in shape:
                    n *= d
                w = local[off:off + n]
                e = (got[off:off + n] - w).abs().max().item()
                sc = w.abs().max().item()
                assert e <= 1e-4 * sc + 1e-8, (mode, name, e, sc)
        # and a whole optimiser step through step(): wait() inside apply_gradients
        loss = model.step(inputs, lengths, mel, lin)
        assert loss == loss and loss < 100
        model.check_status()
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_CHECK_OK")


if __name__ == "__main__":
    main()
