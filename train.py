#!/usr/bin/env python3
"""Training CLI with the reference's flags (train.py:133-164) on the MI355X-native model.

  python3 train.py --ljspeech DIR --model taco2 [--hparams k=v,...] [--restore-step N]
  torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...     (data parallel)

Outputs follow the reference: LOGDIR/RUN/train.log, model.ckpt-STEP (torch.save of a name->tensor
dict whose names mirror the TF scopes), step-%06d-audio.wav every --checkpoint-interval."""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd import parallel  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402
from nspeech_amd.utils import audio  # noqa: E402


class ValueWindow(object):
    """100-step moving average (utils/__init__.py:8-29)."""

    def __init__(self, size=100):
        self.size, self.v = size, []

    def append(self, x):
        self.v = self.v[-(self.size - 1):] + [x]

    @property
    def average(self):
        return sum(self.v) / max(1, len(self.v))


def log(msg, path=None):
    print(msg, flush=True)
    if path:
        with open(path, "a") as f:
            f.write("[%s]  %s\n" % (time.strftime("%Y-%m-%d %H:%M:%S"), msg))


class CheckpointSaver(object):
    """tf.train.Saver(max_to_keep=5, keep_checkpoint_every_n_hours=2) of the reference (train.py:60): the newest
    `max_to_keep` files stay; when an older one falls out of that window it is deleted - unless it was written later
    than the next keep-forever mark, in which case it stays for good and the mark moves on by `keep_every_n_hours`
    (the mark starts that far after the saver was created)."""

    def __init__(self, log_dir, max_to_keep=5, keep_every_n_hours=2.0, clock=time.time):
        self.log_dir, self.max_to_keep, self.every = log_dir, max_to_keep, keep_every_n_hours * 3600.0
        self.clock = clock
        self.recent = []                                   # [(path, time written)], oldest first
        self.next_keep_time = clock() + self.every

    def save(self, model, step):
        path = os.path.join(self.log_dir, "model.ckpt-%d" % step)
        torch.save(model.state_dict(), path)
        with open(os.path.join(self.log_dir, "checkpoint"), "w") as f:
            # the state file of tf.train.Saver, for tools that look up the newest step; the comment line (legal in a
            # text proto) says what the file it names is, since it is not a TensorBundle
            f.write('# model.ckpt-%d is a torch.save dictionary (nspeech_amd); tf_bundle.export_model writes TF bundles\n' % step)
            f.write('model_checkpoint_path: "model.ckpt-%d"\n' % step)
        self.recent = [r for r in self.recent if r[0] != path] + [(path, self.clock())]
        while len(self.recent) > self.max_to_keep:
            old, written = self.recent.pop(0)
            if written > self.next_keep_time:
                self.next_keep_time += self.every         # kept for good
            elif os.path.exists(old):
                os.remove(old)
        return path


def train_step(model, batches, world=1):
    """One iteration of the training loop = the reference's sess.run([global_step, loss, optimize]) (train.py:80): take
    the next batch (targets already on their way to the GPU, see DeviceStager), run the step, read the loss back.
    Returns the loss (the mean over the ranks when data-parallel, so that every rank takes the same abort decision)."""
    inputs, lengths, mel, lin = batches.next_batch()
    loss = model.step(inputs, lengths, mel, lin, speaker_ids=batches.speaker_ids)
    if world > 1:
        t = torch.tensor([loss], device="cuda")
        torch.distributed.all_reduce(t)
        loss = float(t.item()) / world
    return loss


class LossPipeline(object):
    """The training loop with the loss read-back one step behind the launches (default; --sync-loss turns it off): step
    k's launches are issued, then step k-1's loss - already on its way into pinned memory - is waited for, logged and
    checked.  Every step's loss is read and the explosion check of train.py:87-89 sees every value, one step later; the
    GPU never waits for the host between two steps (measured at the benchmark shape: 22.2 ms per step with a
    synchronous read-back against 19.9 ms of device time, bench.py `e2e`)."""

    def __init__(self, model, batches, world=1):
        self.model, self.batches, self.world = model, batches, world
        self.pending = None

    def _finish(self, handle):
        loss = self.model.losses_finish(handle)[0]
        if self.world > 1:      # every rank must agree on the abort decision
            t = torch.tensor([loss], device="cuda")
            torch.distributed.all_reduce(t)
            loss = float(t.item()) / self.world
        return handle["step"], loss, handle["lr"]

    def step(self):
        """Issue one step; returns (step, loss, learning rate) of the step BEFORE it, or None on the first call."""
        inputs, lengths, mel, lin = self.batches.next_batch()
        handle = self.model.step(inputs, lengths, mel, lin, speaker_ids=self.batches.speaker_ids, read_loss="async")
        done = self._finish(self.pending) if self.pending is not None else None
        self.pending = handle
        return done

    def drain(self):
        done = self._finish(self.pending) if self.pending is not None else None
        self.pending = None
        return done


def write_summary(path, step, stats):
    """One JSON object per summary step in LOGDIR/events.jsonl: what the reference hands tf.summary.FileWriter
    (train.py:63,91-93; tacotron2.py:163-188) as scalars - loss, loss_mel, loss_linear, learning_rate,
    max_gradient_norm - plus the per-variable gradient norms and min / max / mean / std of the four value histograms."""
    import json
    with open(path, "a") as f:
        f.write(json.dumps(dict(step=int(step), wall_time=time.time(), **stats)) + "\n")


def train(log_dir, args):
    rank, local, world = parallel.init_distributed()
    torch.cuda.set_device(local)
    hp = hparams_mod.get_hparams()
    logf = os.path.join(log_dir, "train.log") if rank == 0 else None
    log("Checkpoint path: %s" % os.path.join(log_dir, "model.ckpt"), logf)
    log(hparams_mod.debug_string(hp), logf)
    from nspeech_amd.datasets.datafeeder import DataFeeder, DeviceStager
    cmu = None
    if getattr(hp, "use_cmudict", False):       # datafeeder.py:96-108 (commented out in the reference)
        from nspeech_amd.utils.text import cmudict
        cmudict_path = os.path.join(args.ljspeech, "cmudict-0.7b")
        if not os.path.isfile(cmudict_path):
            raise Exception("If use_cmudict=True, you must download "
                            "http://svn.code.sf.net/p/cmusphinx/code/trunk/cmudict/cmudict-0.7b to %s" % cmudict_path)
        cmu = cmudict.CMUDict(cmudict_path, keep_ambiguous=False)
        log("Loaded CMUDict with %d unambiguous entries" % len(cmu), logf)
    # shared seed: every rank walks the same item order and keeps its round-robin share of each sorted group
    feeder = DataFeeder(hp, ljspeech=args.ljspeech, vctk=args.vctk, librispeech=args.librispeech, seed=1234, rank=rank,
                        world=world, cmudict=cmu, pinned=True,
                        device_cache=(args.feature_cache == "device")).start()
    hp.num_speakers = len(feeder.speaker2id)        # train.py:45
    log("Loaded %d different speaker(s)" % hp.num_speakers, logf)
    model = create_model(args.model, hp, device="cuda:%d" % local, dtype=args.precision, world_size=world)
    step0 = 0
    if args.restore_step:
        path = "%s-%d" % (os.path.join(log_dir, "model.ckpt"), args.restore_step)
        from nspeech_amd.utils import tf_bundle
        if tf_bundle.is_bundle(path):          # a TensorFlow checkpoint of the reference (Adam slots taken when complete)
            tf_bundle.load_into_model(model, path)
        else:
            model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
        step0 = model.global_step
        log("Resuming from checkpoint: %s" % path, logf)
    if world > 1:
        parallel.broadcast_parameters(model, 0)
        model.reducer = parallel.make_reducer(model)
    model.add_loss()
    model.add_optimizer(step0)
    model.add_stats()
    time_window, loss_window = ValueWindow(100), ValueWindow(100)
    saver = CheckpointSaver(log_dir)
    batches = DeviceStager(feeder, "cuda:%d" % local)      # pinned H2D of batch k+1 on a copy stream under step k
    events = os.path.join(log_dir, "events.jsonl") if rank == 0 else None
    state = dict(paths=None, t_prev=time.time())

    def report(step, loss):
        """Log line + explosion check of one finished step (train.py:82-89)."""
        paths = getattr(model, "last_paths", None)
        if paths and paths != state["paths"]:       # which kernel family ran each recurrence (a batch shape that falls
            state["paths"] = dict(paths)            # off the persistent kernels runs ~2x slower: say so, once per change)
            log("Recurrence kernels: %s" % ", ".join("%s=%s" % kv for kv in sorted(paths.items())), logf)
        now = time.time()
        time_window.append(now - state["t_prev"])
        state["t_prev"] = now
        loss_window.append(loss)
        frames = model.mel_targets.shape[0] * model.mel_targets.shape[1] * world
        log("Step %-7d [%.03f sec/step, loss=%.05f, avg_loss=%.05f, %.0f mel_frames/s]" %
            (step, time_window.average, loss, loss_window.average, frames / time_window.average), logf)
        if loss > 100 or math.isnan(loss):          # train.py:87-89
            log("Loss exploded to %.05f at step %d!" % (loss, step), logf)
            raise Exception("Loss Exploded")

    pipe = None if args.sync_loss else LossPipeline(model, batches, world)
    while args.max_steps is None or model.global_step < args.max_steps:
        nxt = model.global_step + 1
        due_summary = bool(args.summary_interval) and nxt % args.summary_interval == 0
        due_ckpt = nxt % args.checkpoint_interval == 0
        if pipe is not None and not (due_summary or due_ckpt):
            done = pipe.step()                      # issue step `nxt`, collect the loss of the step before it
            if done is not None:
                report(done[0], done[1])
            continue
        # a step whose summary / checkpoint is due runs synchronously: the files describe the model after exactly `nxt`
        # updates (every rank takes this branch at the same step - the loss all-reduces stay in the same order)
        if pipe is not None:
            done = pipe.drain()
            if done is not None:
                report(done[0], done[1])
        loss = train_step(model, batches, world)
        step = model.global_step
        report(step, loss)
        if rank == 0 and due_summary:                                                   # train.py:91-93
            log("Writing summary at step: %d" % step, logf)
            write_summary(events, step, model.stats())
        if rank == 0 and due_ckpt:
            path = saver.save(model, step)
            log("Saved checkpoint %s" % path, logf)
            wav = audio.inv_preemphasis(model.audio[0].cpu().numpy())      # train.py:100-107 fetches model.audio[0]
            audio.save_wav(wav[:audio.find_endpoint(wav)], os.path.join(log_dir, "step-%06d-audio.wav" % step))
    if pipe is not None:
        done = pipe.drain()
        if done is not None:
            report(done[0], done[1])
    return model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-dir", "--log_dir", default=os.path.expanduser("~/nspeech-logs"))
    ap.add_argument("--input", default="training/train.txt")
    ap.add_argument("--vctk", default=None)
    ap.add_argument("--ljspeech", default=None)
    ap.add_argument("--librispeech", default=None)
    ap.add_argument("--model", default="taco2")
    ap.add_argument("--name", default=None)
    ap.add_argument("--hparams", default="")
    ap.add_argument("--restore-step", "--restore_step", type=int, default=None)
    ap.add_argument("--summary-interval", "--summary_interval", type=int, default=100)
    ap.add_argument("--checkpoint-interval", "--checkpoint_interval", type=int, default=1000)
    ap.add_argument("--slack-url", "--slack_url", default=None)
    ap.add_argument("--tf-log-level", "--tf_log_level", type=int, default=1)
    ap.add_argument("--git", action="store_true")
    ap.add_argument("--gpu", default="0")
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--precision", default="mixed", choices=["mixed", "bf16", "bf16x3", "fp32"])
    ap.add_argument("--max-steps", "--max_steps", type=int, default=None)
    ap.add_argument("--sync-loss", "--sync_loss", action="store_true",
                    help="wait for every step's loss before issuing the next step (default: the read-back runs one step "
                         "behind the launches, see LossPipeline)")
    ap.add_argument("--feature-cache", "--feature_cache", default="device", choices=["device", "host"],
                    help="where the feeder keeps the spectrogram features of the corpus: 'device' = in HBM, batches "
                         "assembled on the GPU (all of LJSpeech is 30.5 GB of float32); 'host' = in RAM as the reference "
                         "does (datafeeder.py:165-176), uploaded through pinned buffers on a copy stream")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", args.gpu)
    run_name = args.name or args.model
    log_dir = os.path.join(args.log_dir, "logs-%s" % run_name)
    os.makedirs(log_dir, exist_ok=True)
    hp = hparams_mod.load(args.model)
    hp.parse(args.hparams)
    train(log_dir, args)


if __name__ == "__main__":
    main()
