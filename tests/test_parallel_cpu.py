"""world_size-2 gloo rehearsal of the data-parallel path on CPU: bucket layout, asynchronous
bucketed all-reduce, parameter broadcast semantics.  (The HIP kernels need a GPU; what is tested
here is the host logic that bench.py / train.py run per rank.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nspeech_amd import hparams as hparams_mod
from nspeech_amd import parallel
from nspeech_amd.models import params as P


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bucket_ranges_tile_the_flat_buffer():
    hp = hparams_mod.load("taco2")
    lay, _ = P.taco2_layout(hp, 149)
    b = parallel.bucket_ranges(lay)
    assert [n for n, _, _ in b] == ["head", "postnet", "decoder", "encoder"]   # backward completion order
    covered = sum(hi - lo for _, lo, hi in b)
    assert covered == lay.size
    sizes = {n: hi - lo for n, lo, hi in b}
    # 7.56 M (head) / 5.50 M (postnet) / 16.93 M (decoder) / 4.89 M (encoder) parameters
    assert abs(sizes["head"] - 7556097) < 2000 and abs(sizes["postnet"] - 5496400) < 2000
    assert abs(sizes["decoder"] - 16927900) < 2000 and abs(sizes["encoder"] - 4894464) < 2000
    # multi-speaker layouts (tacotron2.py:40-47, rnn_wrappers.py:28-30) still tile: the speaker table travels with the
    # encoder bucket, its projection and the widened attention LSTM kernel with the decoder bucket
    hp.num_speakers = 109
    lay2, _ = P.taco2_layout(hp, 149)
    b2 = parallel.bucket_ranges(lay2)
    assert sum(hi - lo for _, lo, hi in b2) == lay2.size
    s2 = {n: hi - lo for n, lo, hi in b2}
    assert s2["head"] == sizes["head"] and s2["postnet"] == sizes["postnet"]
    assert s2["encoder"] - sizes["encoder"] == 109 * 16
    assert s2["decoder"] - sizes["decoder"] == 16 * 128 + 128 + 128 * 4 * hp.attention_dim


def _worker(rank, world, port, size, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, _, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    g = torch.arange(size, dtype=torch.float32) * (rank + 1)
    buckets = [("head", 600, size), ("postnet", 400, 600), ("decoder", 100, 400), ("encoder", 0, 100)]
    red = parallel.GradReducer(g, buckets)
    for name, _, _ in buckets:          # the order the backward pass releases them
        red.bucket_ready(name)
    red.wait()
    expect = torch.arange(size, dtype=torch.float32) * sum(range(1, world + 1))
    ok = torch.equal(g, expect)
    # parameter broadcast: every rank ends with rank 0's values
    p = torch.full((16,), float(rank + 7))
    dist.broadcast(p, 0)
    ok = ok and bool((p == 7).all())
    out[rank] = ok
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    world, size = 2, 1000
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, size, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_reducer_refuses_a_step_that_skipped_a_bucket():
    g = torch.zeros(100)
    red = parallel.GradReducer(g, [("head", 50, 100), ("encoder", 0, 50)])
    red.bucket_ready("head")
    with __import__("pytest").raises(RuntimeError, match="exactly once"):
        red.wait()                       # 'encoder' never released
    red.bucket_ready("head")
    red.bucket_ready("encoder")
    red.wait()                           # a complete step passes, and the counts start afresh
    red.bucket_ready("head")
    red.bucket_ready("head")
    red.bucket_ready("encoder")
    with __import__("pytest").raises(RuntimeError, match="exactly once"):
        red.wait()                       # released twice
    with __import__("pytest").raises(KeyError):
        red.bucket_ready("postnet")


def test_bucket_layout_matches_the_model_family():
    """train.py applies data parallelism to any --model: Tacotron-1 has none of the Tacotron-2 scopes, so it gets one
    whole-buffer bucket released after backward(); asking for the Tacotron-2 buckets on its layout is a clear error."""
    hp1 = hparams_mod.load("taco1")
    lay1, _ = P.taco1_layout(hp1, 149)
    with __import__("pytest").raises(ValueError, match="not a Tacotron-2 layout|outside every bucket"):
        parallel.bucket_ranges(lay1)
    assert parallel.whole_buffer_range(lay1) == [("all", 0, lay1.size)]

    class M1:          # what make_reducer reads off a model
        _BUCKET_AFTER = {"backward": "all"}
        layout = lay1
        flat_g = torch.zeros(lay1.size)
    red = parallel.make_reducer(M1())
    assert list(red.buckets) == ["all"]
    hp2 = hparams_mod.load("taco2")
    lay2, _ = P.taco2_layout(hp2, 149)

    from nspeech_amd.models.tacotron2 import Tacotron2
    # no bucket goes to the collective directly in front of the whole-chip decoder recurrences (the phases
    # behind "postnet_bwd" and "dec_lstm_bwd"); the postnet bucket waits for the end of the attention RNN's backward
    assert "postnet_bwd" not in Tacotron2._BUCKET_AFTER and "dec_lstm_bwd" not in Tacotron2._BUCKET_AFTER
    assert Tacotron2._BUCKET_AFTER["attn_rnn_bwd"] == "postnet"

    class M2:
        _BUCKET_AFTER = Tacotron2._BUCKET_AFTER
        layout = lay2
        flat_g = torch.zeros(lay2.size)
    assert sorted(parallel.make_reducer(M2()).buckets) == ["decoder", "encoder", "head", "postnet"]
