#!/usr/bin/env python3
"""WaveNet training CLI with the reference's flags (train_wavenet.py:19-135) on the MI355X-native models.

  python3 train_wavenet.py --ljspeech DIR [--model wavenet | simple_wavenet] [--hparams sample_size=8000,batch_size=8] ...

--model wavenet (the reference's default, :108) is WaveNetModel with its options - e.g. --hparams gc_channels=32 conditions
every piece on its speaker (the feeder's speaker ids, :40-46; the embedding table is sized from the corpora BEFORE the
model is built - the reference builds the model first, with the yaml's cardinality 0); with the shipped options it is
simple_wavenet.  lc_channels > 0 is refused here: the reference's pieces carry a mel image of receptive_field rows
(WavenetDataFeeder.py:127-135), which no layer's output length equals - its graph does not build either.

LOGDIR/RUN/train.log, model.ckpt-STEP (torch.save of a name -> tensor dict under the TF variable names), a scalars line
in events.jsonl every --summary-interval steps.  The shipped train.yaml has sample_size = 1 (one predicted sample per
piece); pass a larger one for real training."""
import argparse
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402
from train import CheckpointSaver, ValueWindow, log, write_summary  # noqa: E402


def train_wavenet(log_dir, args, hp):
    from nspeech_amd.datasets.wavenet_feeder import WavenetFeeder
    logf = os.path.join(log_dir, "train.log")
    log("Checkpoint path: %s" % os.path.join(log_dir, "model.ckpt"), logf)
    log("Using model: %s" % args.model, logf)
    log(hparams_mod.debug_string(hp), logf)
    from nspeech_amd.models.wavenet import receptive_field
    full = args.model == "wavenet"
    feeder = WavenetFeeder(hp, receptive_field(hp, full), ljspeech=args.ljspeech or None, vctk=args.vctk or None,
                           librispeech=args.librispeech or None, seed=1234)
    log("Loaded data refs for %d examples" % len(feeder.items), logf)
    log("Loaded %d different speaker(s)" % len(feeder.speaker2id), logf)
    hp.num_speakers = len(feeder.speaker2id)            # train_wavenet.py:40-41
    hp.gc_category_cardinality = hp.num_speakers
    model = create_model(args.model, hp, device="cuda:0", dtype=args.precision)
    use_gc = full and (hp.gc_channels or 0) > 0         # :46
    step0 = 0
    if args.restore_step:
        path = "%s-%d" % (os.path.join(log_dir, "model.ckpt"), args.restore_step)
        model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
        step0 = model.global_step
        log("Resuming from checkpoint: %s" % path, logf)
    else:
        log("Starting new training run ", logf)
    model.add_loss(hp.l2_regularization_strength or None)
    model.add_optimizer(step0)
    model.add_stats()
    time_window, loss_window = ValueWindow(100), ValueWindow(100)
    saver = CheckpointSaver(log_dir)
    events = os.path.join(log_dir, "events.jsonl")
    while args.max_steps is None or model.global_step < args.max_steps:
        t0 = time.time()
        batch = feeder.next_batch()
        loss = model.step(batch, feeder.speaker_ids if use_gc else None)        # train_wavenet.py:46-49, 75
        step = model.global_step
        time_window.append(time.time() - t0)
        loss_window.append(loss)
        log("Step %-7d [%.03f sec/step, loss=%.05f, avg_loss=%.05f, queue=%.02f]" % (
            step, time_window.average, loss, loss_window.average, feeder.size / float(feeder.capacity)), logf)
        if loss > 100 or math.isnan(loss):
            log("Loss exploded to %.05f at step %d!" % (loss, step), logf)
            raise Exception("Loss Exploded")
        if args.summary_interval and step % args.summary_interval == 0:
            log("Writing summary at step: %d" % step, logf)
            write_summary(events, step, model.stats())
        if step % args.checkpoint_interval == 0:
            log("Saving checkpoint to: %s" % saver.save(model, step), logf)
    return model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-dir", "--log_dir", default=os.path.expanduser("logs"))
    ap.add_argument("--vctk", default="")
    ap.add_argument("--ljspeech", default="")
    ap.add_argument("--librispeech", default="")
    ap.add_argument("--model", default="wavenet", choices=["wavenet", "simple_wavenet"])       # train_wavenet.py:108
    ap.add_argument("--name", default=None)
    ap.add_argument("--hparams", default="")
    ap.add_argument("--restore-step", "--restore_step", type=int, default=None)
    ap.add_argument("--summary-interval", "--summary_interval", type=int, default=1000)
    ap.add_argument("--checkpoint-interval", "--checkpoint_interval", type=int, default=1000)
    ap.add_argument("--slack-url", "--slack_url", default=None)
    ap.add_argument("--tf-log-level", "--tf_log_level", type=int, default=1)
    ap.add_argument("--git", action="store_true")
    ap.add_argument("--gpu", default=0, type=int)
    ap.add_argument("--threads", default=1, type=int)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--max-steps", "--max_steps", type=int, default=None)
    args = ap.parse_args()
    os.environ.setdefault("HIP_VISIBLE_DEVICES", str(args.gpu))
    run_name = args.name or args.model
    log_dir = os.path.join(args.log_dir, run_name)
    os.makedirs(log_dir, exist_ok=True)
    hp = hparams_mod.load("wavenet")
    hp.parse(args.hparams)
    if (hp.lc_channels or 0) > 0:
        sys.exit("train_wavenet.py: lc_channels > 0 - the feeder's local-condition images have receptive_field rows "
                 "(WavenetDataFeeder.py:127-135) and fit no layer; pass local conditions to WaveNetModel.initialize yourself")
    train_wavenet(log_dir, args, hp)


if __name__ == "__main__":
    main()
