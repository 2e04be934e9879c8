#!/usr/bin/env python3
"""Evaluation CLI (flags of the reference's eval.py:62-76): synthesises the test sentences with a
checkpoint and writes eval-STEP-i.wav next to it."""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.synthesizer import Synthesizer  # noqa: E402
from nspeech_amd.utils import audio  # noqa: E402

sentences = [
    "Scientists at the CERN laboratory say they have discovered a new particle.",
    "There's a way to measure the acute emotional intelligence that has never gone out of style.",
    "President Trump met with other leaders at the Group of 20 conference.",
    "The Senate's bill to repeal and replace the Affordable Care Act is now imperiled.",
    "Generative adversarial network or variational auto-encoder.",
    "The buses aren't the problem, they actually provide a solution.",
]


def get_output_base_path(checkpoint_path):
    base_dir = os.path.dirname(checkpoint_path)
    m = re.compile(r".*?\.ckpt\-([0-9]+)").match(checkpoint_path)
    name = "eval-%d" % int(m.group(1)) if m else "eval"
    return os.path.join(base_dir, name)


def run_eval(args):
    hp = hparams_mod.get_hparams()
    synth = Synthesizer(hp, dtype=args.precision).load(args.checkpoint, args.model)
    base_path = get_output_base_path(args.checkpoint)
    for i, text in enumerate(sentences):
        path = "%s-%d.wav" % (base_path, i)
        print("Synthesizing: %s" % path)
        wav, mel, lin = synth.synthesize(text, args.speaker)
        audio.save_wav(wav, path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--model", default="taco2")
    ap.add_argument("--hparams", default="")
    ap.add_argument("--gpu", default="0")
    ap.add_argument("--speaker", type=int, default=0)
    ap.add_argument("--precision", default="mixed")
    args = ap.parse_args()
    os.environ.setdefault("HIP_VISIBLE_DEVICES", args.gpu)
    hp = hparams_mod.load(args.model)
    hp.parse(args.hparams)
    run_eval(args)


if __name__ == "__main__":
    main()
