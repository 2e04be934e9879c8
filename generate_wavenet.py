#!/usr/bin/env python3
"""Waveform generation from a simple_wavenet checkpoint, the reference's generate_wavenet.py surface (:48-224): positional
checkpoint, --samples --temperature --wav_out_path --save_every --fast_generation --wav_seed.

The reference script builds WaveNetModel and needs a wavenet_params.json that the repository does not ship
(generate_wavenet.py:21,51); here the network is wavenet.yaml's (+ --hparams; --gc_channels / --gc_cardinality / --gc_id
for a speaker-conditioned checkpoint of train_wavenet.py --model wavenet, :214-231), restored from
train_wavenet.py's model.ckpt-<step>.  --fast_generation true (default) at temperature 1.0 = the persistent incremental
generator on the GPU (one workgroup per waveform; float64 softmax and inverse-CDF draw in the kernel).  Any other
temperature, or --fast_generation false, takes the full-window path of generate_wavenet.py:104-142: predict_proba over the
last receptive_field samples, temperature scaling and the draw on the host, including the reference's own consistency
check at temperature 1.0 (:133-138, scaled == unscaled)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402
from nspeech_amd.models.wavenet import mu_law_decode, mu_law_encode  # noqa: E402
from nspeech_amd.utils import audio  # noqa: E402

SAMPLES = 16000
TEMPERATURE = 1.0
SILENCE_THRESHOLD = 0.1
SAMPLE_RATE = 16000          # the reference reads it from wavenet_params.json; WaveNet runs on 16 kHz mu-law audio


def write_wav(waveform, sample_rate, filename):
    import wave
    y = np.clip(np.asarray(waveform, np.float64), -1.0, 1.0)
    with wave.open(filename, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sample_rate)
        f.writeframes((y * 32767.0).astype("<i2").tobytes())
    print("Updated wav file at {}".format(filename))


def create_seed(filename, quantization_channels, window_size, silence_threshold=SILENCE_THRESHOLD):
    """generate_wavenet.py:32-45: load, trim silence, mu-law encode, keep at most window_size samples."""
    from nspeech_amd.datasets.wavenet_feeder import trim_silence
    wav = trim_silence(np.asarray(audio.load_wav(filename), np.float32), silence_threshold)
    return mu_law_encode(wav[None], quantization_channels)[0][:window_size]


def scale_prediction(prediction, temperature):
    """generate_wavenet.py:124-130."""
    with np.errstate(divide="ignore"):
        scaled = np.log(prediction) / temperature
        scaled = scaled - np.logaddexp.reduce(scaled)
        return np.exp(scaled)


def main(args):
    hp = hparams_mod.load("wavenet")
    hp.parse(args.hparams)
    gc = None
    if args.gc_channels is not None:                # generate_wavenet.py:221-231: the speaker whose voice is drawn
        hp.gc_channels, hp.gc_category_cardinality = args.gc_channels, args.gc_cardinality
        gc = np.asarray([args.gc_id])
    full = bool(hp.use_biases or hp.scalar_input or hp.gc_channels or hp.lc_channels)
    net = create_model("wavenet" if full else "simple_wavenet", hp, device="cuda:0", dtype=args.precision)
    cond = dict(global_conditions=gc) if full else {}
    cond1 = dict(global_condition=None if gc is None else gc[0]) if full else {}
    print("Restoring model from {}".format(args.checkpoint))
    net.load_state_dict(torch.load(args.checkpoint, map_location="cpu", weights_only=True))
    q, rf = hp.quantization_channels, net.rf
    rng = np.random.default_rng(args.seed)
    if args.wav_seed:
        waveform = create_seed(args.wav_seed, q, rf).tolist()
        if len(waveform) < rf:                      # the generator wants a full window: silence in front
            waveform = [q // 2] * (rf - len(waveform)) + waveform
    else:                                           # silence with a single random sample at the end (:84-86)
        waveform = [q // 2] * (rf - 1) + [int(rng.integers(q))]
    fast = args.fast_generation and args.temperature == 1.0
    if fast:
        done = 0
        while done < args.samples:                  # in chunks, so that --save_every can write partial results
            n = min(args.samples - done, args.save_every or args.samples)
            ids = net.generate(np.asarray(waveform[-rf:], np.int32), n, uniforms=rng.random((1, n)), **cond)
            waveform.extend(int(x) for x in ids[0, rf:].cpu().numpy())
            done += n
            print("Sample {:3<d}/{:3<d}".format(done, args.samples), end="\r")
            if args.wav_out_path and args.save_every and done < args.samples:
                write_wav(mu_law_decode(np.asarray(waveform), q), SAMPLE_RATE, args.wav_out_path)
    else:
        for step in range(args.samples):
            window = waveform[-rf:] if len(waveform) > rf else waveform
            prediction = net.predict_proba(window, **cond1).double().cpu().numpy()
            scaled = scale_prediction(prediction, args.temperature)
            if args.temperature == 1.0:             # the reference's own check (:133-138)
                np.testing.assert_allclose(prediction, scaled, atol=1e-5,
                                           err_msg="Prediction scaling at temperature=1.0 is not working as intended.")
            waveform.append(int(rng.choice(np.arange(q), p=scaled / scaled.sum())))
            if (step + 1) % 100 == 0:
                print("Sample {:3<d}/{:3<d}".format(step + 1, args.samples), end="\r")
            if args.wav_out_path and args.save_every and (step + 1) % args.save_every == 0:
                write_wav(mu_law_decode(np.asarray(waveform), q), SAMPLE_RATE, args.wav_out_path)
    print()
    if args.wav_out_path:
        write_wav(mu_law_decode(np.asarray(waveform), q), SAMPLE_RATE, args.wav_out_path)
    print("Finished generating.")
    return waveform


if __name__ == "__main__":
    def _str_to_bool(s):
        if s.lower() not in ["true", "false"]:
            raise ValueError("Argument needs to be a boolean, got {}".format(s))
        return {"true": True, "false": False}[s.lower()]

    def _ensure_positive_float(f):
        if float(f) < 0:
            raise argparse.ArgumentTypeError("Argument must be greater than zero")
        return float(f)

    parser = argparse.ArgumentParser(description="WaveNet generation script")
    parser.add_argument("checkpoint", type=str, help="Which model checkpoint to generate from")
    parser.add_argument("--samples", type=int, default=SAMPLES)
    parser.add_argument("--temperature", type=_ensure_positive_float, default=TEMPERATURE)
    parser.add_argument("--logdir", type=str, default="./logdir")
    parser.add_argument("--wavenet_params", type=str, default=None, help="ignored: the network is wavenet.yaml's")
    parser.add_argument("--wav_out_path", type=str, default=None)
    parser.add_argument("--save_every", type=int, default=None)
    parser.add_argument("--fast_generation", type=_str_to_bool, default=True)
    parser.add_argument("--wav_seed", type=str, default=None)
    parser.add_argument("--gc_channels", type=int, default=None)
    parser.add_argument("--gc_cardinality", type=int, default=None)
    parser.add_argument("--gc_id", type=int, default=None)
    parser.add_argument("--hparams", default="")
    parser.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    parser.add_argument("--seed", type=int, default=0)
    a = parser.parse_args()
    if a.gc_channels is not None:                   # generate_wavenet.py:221-231
        if a.gc_cardinality is None:
            raise ValueError("Globally conditioning but gc_cardinality not specified. Use --gc_cardinality=377 for full "
                             "VCTK corpus.")
        if a.gc_id is None:
            raise ValueError("Globally conditioning, but global condition was not specified. Use --gc_id to specify global "
                             "condition.")
    main(a)
