"""Data parallelism for the Tacotron training step: one process per GPU, torch.distributed with
the 'nccl' backend (= RCCL over xGMI on MI355X), gradients summed with bucketed, asynchronous
all-reduces that overlap the rest of the backward pass.

The reference has no multi-GPU code; this is the north_star's data-parallel path.
The flat gradient buffer is laid out in forward order, so the four buckets - expand+linear head,
postnet, decoder+attention, encoder+embedding - are contiguous slices that become final in exactly
that order during the backward pass; each is handed to RCCL as soon as the kernels that write it
have been enqueued (the collective waits on the compute stream, then runs on RCCL's own stream
while the decoder's backward-through-time keeps the compute stream busy).
clip_by_global_norm and Adam then run on the reduced sum with grad_scale = 1/world.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, local, world


def bucket_ranges(layout):
    """Contiguous [lo, hi) slices of the flat buffer in the order the backward pass finishes them."""
    groups = [("head", ("expand/", "dense/")), ("postnet", ("decoder_postnet/",)),
              ("decoder", ("decoder/", "attention_decoder/")), ("encoder", ("encoder/", "embedding/", "speaker/"))]
    out = []
    for gname, prefixes in groups:
        offs = [(o, o + ((int(_numel(s)) + 7) // 8) * 8) for n, (o, s) in layout.entries.items()
                if any(n.startswith(p) for p in prefixes)]
        lo, hi = min(a for a, _ in offs), max(b for _, b in offs)
        out.append((gname, lo, hi))
    # the groups must tile the buffer without interleaving
    spans = sorted((lo, hi) for _, lo, hi in out)
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0, "parameter groups interleave in the flat buffer"
    return out


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


class GradReducer(object):
    """Asynchronous bucketed sum-all-reduce of a flat gradient buffer."""

    def __init__(self, flat_g, buckets, group=None):
        self.flat_g = flat_g
        self.buckets = {name: (lo, hi) for name, lo, hi in buckets}
        self.group = group
        self.pending = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def bucket_ready(self, name):
        if self.world == 1:
            return
        lo, hi = self.buckets[name]
        self.pending.append(dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []


def broadcast_parameters(model, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(model.flat_p, src)
        dist.broadcast(model.flat_stats, src)
        model.refresh_shadows(full=True)
