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
    """Contiguous [lo, hi) slices of the flat buffer in the order the Tacotron-2 backward pass finishes them."""
    groups = [("head", ("expand/", "dense/")), ("postnet", ("decoder_postnet/",)),
              ("decoder", ("decoder/", "attention_decoder/")), ("encoder", ("encoder/", "embedding/", "speaker/"))]
    out, claimed = [], set()
    for gname, prefixes in groups:
        names = [n for n in layout.entries if any(n.startswith(p) for p in prefixes)]
        if not names:
            raise ValueError("bucket_ranges: no parameter starts with %s (group '%s'): this is not a Tacotron-2 layout - "
                             "use whole_buffer_range() for other models" % ("|".join(prefixes), gname))
        claimed.update(names)
        offs = [(layout.entries[n][0], layout.entries[n][0] + ((int(_numel(layout.entries[n][1])) + 7) // 8) * 8)
                for n in names]
        out.append((gname, min(a for a, _ in offs), max(b for _, b in offs)))
    left = sorted(set(layout.entries) - claimed)
    if left:
        raise ValueError("bucket_ranges: parameters outside every bucket (they would never be reduced): %s" % left[:8])
    # the groups must tile the buffer without interleaving
    spans = sorted((lo, hi) for _, lo, hi in out)
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        if a1 > b0:
            raise ValueError("bucket_ranges: parameter groups interleave in the flat buffer")
    return out


def whole_buffer_range(layout):
    """One bucket for models whose backward pass has no per-group hand-off points (Tacotron-1): reduced after backward()."""
    return [("all", 0, layout.size)]


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


class GradReducer(object):
    """Asynchronous bucketed sum-all-reduce of a flat gradient buffer.

    bucket_ready(name) is called by the model when the last kernel writing that slice has been enqueued; wait() is
    called before the optimiser.  wait() REFUSES to return unless every bucket went to the collective exactly once
    since the previous wait(): a bucket the model never released would leave the ranks with different gradients
    without any other symptom.  `force` runs the collectives even in a one-rank group (tests: RCCL load, stream
    hand-off and completion on a one-GPU box)."""

    def __init__(self, flat_g, buckets, group=None, force=False):
        self.flat_g = flat_g
        self.buckets = {name: (lo, hi) for name, lo, hi in buckets}
        self.group = group
        self.pending = []
        self.issued = {}
        self.joined = True            # no step in flight
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())

    def bucket_ready(self, name):
        if name not in self.buckets:
            raise KeyError("GradReducer: unknown bucket '%s' (have %s)" % (name, sorted(self.buckets)))
        self.issued[name] = self.issued.get(name, 0) + 1
        self.joined = False
        if not self.active:
            return
        lo, hi = self.buckets[name]
        self.pending.append(dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        if self.joined:               # nothing released since the last complete step (a second wait() is harmless)
            return
        self.joined = True
        bad = {n: self.issued.get(n, 0) for n in self.buckets if self.issued.get(n, 0) != 1}
        self.issued = {}
        pending, self.pending = self.pending, []
        for w in pending:
            w.wait()
        if bad:
            raise RuntimeError("GradReducer: every bucket must be released exactly once per step; release counts %s "
                               "(the model's backward pass and the bucket list do not match)" % bad)


def make_reducer(model, group=None, force=False):
    """The reducer that matches the model's backward pass: the model names its hand-off points in _BUCKET_AFTER."""
    names = set(model._BUCKET_AFTER.values())
    buckets = whole_buffer_range(model.layout) if names == {"all"} else bucket_ranges(model.layout)
    if set(n for n, _, _ in buckets) != names:
        raise ValueError("make_reducer: %s releases buckets %s but the layout gives %s"
                         % (type(model).__name__, sorted(names), [n for n, _, _ in buckets]))
    return GradReducer(model.flat_g, buckets, group=group, force=force)


def broadcast_parameters(model, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(model.flat_p, src)
        dist.broadcast(model.flat_stats, src)
        model.refresh_shadows(full=True)
