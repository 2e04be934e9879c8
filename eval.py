#!/usr/bin/env python3
"""Synthesis from a checkpoint, the reference's eval.py surface (eval.py:23-27,62-76): flags --checkpoint --model
--hparams --gpu --speaker; one `eval-<step>-<i>.wav` per test sentence next to the checkpoint.  The checkpoint may be
this build's `model.ckpt-<step>` file or a TensorFlow checkpoint prefix (nspeech_amd/utils/tf_bundle.py)."""
import argparse
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.synthesizer import Synthesizer  # noqa: E402
from nspeech_amd.utils import audio  # noqa: E402

# exercises the text front end: abbreviations, digits, an apostrophe, a question, an ARPAbet span
TEST_SENTENCES = (
    "Dr. Jones arrived at 9 o'clock with 3 colleagues.",
    "The first measurement took 12.5 milliseconds; the second one took longer.",
    "It's the spectrogram, not the waveform, that the network predicts.",
    "Could you turn left on {HH AW1 S S T AH0 N} Street, please?",
    "Mr. Smith paid $20 for the recording on May 4th.",
    "Wave after wave reached the shore while the gulls kept calling.",
)


def output_stem(checkpoint_path):
    """.../model.ckpt-1234 -> .../eval-1234 (eval.py:23-27 keys the output names on the step in the file name)."""
    step = re.search(r"\.ckpt-(\d+)", os.path.basename(checkpoint_path))
    return os.path.join(os.path.dirname(checkpoint_path), "eval-%d" % int(step.group(1)) if step else "eval")


def synthesize_all(checkpoint, model_name, speaker, precision):
    synth = Synthesizer(hparams_mod.get_hparams(), dtype=precision).load(checkpoint, model_name)
    stem = output_stem(checkpoint)
    for index, sentence in enumerate(TEST_SENTENCES):
        target = "%s-%d.wav" % (stem, index)
        print("Synthesizing: %s" % target)
        wav, _, _ = synth.synthesize(sentence, speaker)
        audio.save_wav(wav, target)


if __name__ == "__main__":
    cli = argparse.ArgumentParser(description=__doc__)
    cli.add_argument("--checkpoint", required=True, help="model.ckpt-<step> (torch file) or a TF checkpoint prefix")
    cli.add_argument("--model", default="taco2")
    cli.add_argument("--hparams", default="", help="comma separated name=value overrides")
    cli.add_argument("--gpu", default="0")
    cli.add_argument("--speaker", type=int, default=0)
    cli.add_argument("--precision", default="mixed", choices=["mixed", "bf16", "bf16x3", "fp32"])
    opts = cli.parse_args()
    os.environ.setdefault("HIP_VISIBLE_DEVICES", opts.gpu)
    hparams_mod.load(opts.model).parse(opts.hparams)
    synthesize_all(opts.checkpoint, opts.model, opts.speaker, opts.precision)
