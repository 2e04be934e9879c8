"""train.py on a tiny synthetic LJSpeech-layout corpus (metadata.csv + wavs/): features on the GPU,
3 optimiser steps, a checkpoint, resume, and eval.py synthesis from that checkpoint."""
import os
import subprocess
import sys
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ("embedding_dim=32,encoder_conv_channels=64,encoder_lstm_units=32,attention_dim=64,decoder_lstm_units=64,"
         "postnet_conv_channels=64,expand_conv_channels=64,expand_lstm_units=32,batch_size=2,batch_group_size=2,max_iters=60")


def _corpus(tmp):
    os.makedirs(os.path.join(tmp, "wavs"))
    rng = np.random.default_rng(0)
    lines = []
    for i, text in enumerate(["Hello world.", "Dr. Smith met Mr. Jones!", "A short one.", "The quick brown fox jumps."]):
        L = int(20000 * rng.uniform(0.5, 0.9))
        t = np.arange(L) / 20000.0
        y = 0.5 * np.sin(2 * np.pi * 180 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 3 * t)) + rng.normal(0, 0.01, L)
        with wave.open(os.path.join(tmp, "wavs", "utt%d.wav" % i), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(20000)
            f.writeframes((np.clip(y, -1, 1) * 32767).astype("<i2").tobytes())
        lines.append("utt%d|%s|%s" % (i, text, text))
    with open(os.path.join(tmp, "metadata.csv"), "w") as f:
        f.write("\n".join(lines) + "\n")


def test_train_checkpoint_resume_eval(tmp_path):
    data = str(tmp_path / "lj")
    os.makedirs(data)
    _corpus(data)
    logs = str(tmp_path / "logs")
    base = [sys.executable, os.path.join(ROOT, "train.py"), "--ljspeech", data, "--model", "taco2", "--log_dir", logs,
            "--hparams", SMALL, "--checkpoint_interval", "2", "--precision", "bf16", "--summary-interval", "1"]
    r = subprocess.run(base + ["--max_steps", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    run = os.path.join(logs, "logs-taco2")
    assert os.path.exists(os.path.join(run, "model.ckpt-2")) and os.path.exists(os.path.join(run, "train.log"))
    assert os.path.exists(os.path.join(run, "step-000002-audio.wav"))
    # --summary-interval (train.py:91-93 / tacotron2.py:163-188): one scalars line per summary step
    import json
    ev = [json.loads(l) for l in open(os.path.join(run, "events.jsonl"))]
    assert [e["step"] for e in ev] == [1, 2]
    for e in ev:
        assert set(("loss", "loss_mel", "loss_linear", "learning_rate", "max_gradient_norm")) <= set(e)
        assert abs(e["loss"] - (e["loss_mel"] + e["loss_linear"])) < 1e-5 * e["loss"]
        assert e["max_gradient_norm"] == max(e["gradient_norm"].values()) > 0
        h = e["histograms"]
        assert set(h) == {"mel_outputs", "linear_outputs", "mel_targets", "linear_targets"}
        assert 0.0 <= h["mel_targets"]["min"] <= h["mel_targets"]["mean"] <= h["mel_targets"]["max"] <= 1.0
    r = subprocess.run(base + ["--max-steps", "3", "--restore-step", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Resuming from checkpoint" in r.stdout and "Step 3 " in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "eval.py"), "--checkpoint", os.path.join(run, "model.ckpt-2"),
                        "--model", "taco2", "--hparams", SMALL, "--precision", "bf16"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(os.path.join(run, "eval-2-0.wav"))


def test_taco1_config1_plumbing(tmp_path):
    """BASELINE configs[0]: --model taco1, 4 utterances, batch_size=2, outputs_per_step=5 (run on the GPU
    here: the product path has no CPU fallback): 3 steps, a checkpoint, one eval synthesis."""
    data = str(tmp_path / "lj")
    os.makedirs(data)
    _corpus(data)
    logs = str(tmp_path / "logs")
    small = ("embedding_dim=32,encoder_prenet=[32,128],encoder_cbhg_banks=4,attention_dim=64,decoder_dim=64,"
             "post_cbhg_banks=3,post_cbhg_bank_sizes=[64],batch_size=2,batch_group_size=2,outputs_per_step=5,max_iters=40")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--ljspeech", data, "--model", "taco1",
                        "--log_dir", logs, "--hparams", small, "--checkpoint_interval", "3", "--max_steps", "3",
                        "--precision", "bf16"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    run = os.path.join(logs, "logs-taco1")
    assert os.path.exists(os.path.join(run, "model.ckpt-3")) and "Step 3 " in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "eval.py"), "--checkpoint", os.path.join(run, "model.ckpt-3"),
                        "--model", "taco1", "--hparams", small, "--precision", "bf16"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(os.path.join(run, "eval-3-0.wav"))


def test_taco1_config1_at_the_shipped_widths(tmp_path):
    """BASELINE configs[0] with hparams/taco1.yaml as it ships (16-bank / 8-bank CBHG, 256-unit GRUs, 1025 bins): the
    persistent Tacotron-1 kernels (GRU recurrences, attention clusters) under train.py's own loop - feeder with silence
    trimming, checkpoint - then eval.py on the checkpoint.  The paths train.py logs must name the persistent forms."""
    data = str(tmp_path / "lj")
    os.makedirs(data)
    _corpus(data)
    logs = str(tmp_path / "logs")
    hps = "batch_size=2,batch_group_size=2,outputs_per_step=5,max_iters=12"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--ljspeech", data, "--model", "taco1",
                        "--log_dir", logs, "--hparams", hps, "--checkpoint_interval", "3", "--max_steps", "3",
                        "--precision", "mixed"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "attn:fwd=cluster" in r.stdout and "post_gru:bwd=seq" in r.stdout and "gru_1:fwd=seq" in r.stdout, r.stdout[-3000:]
    run = os.path.join(logs, "logs-taco1")
    assert os.path.exists(os.path.join(run, "model.ckpt-3")) and "Step 3 " in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "eval.py"), "--checkpoint", os.path.join(run, "model.ckpt-3"),
                        "--model", "taco1", "--hparams", hps, "--precision", "mixed"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(os.path.join(run, "eval-3-0.wav"))


def test_multi_speaker_train_and_eval(tmp_path):
    """LJSpeech + a VCTK-layout corpus (wav48/pNNN/*.wav + txt/pNNN/*.txt, corpus/vctk.py:11-20) give
    three speakers; train.py sizes the speaker table from the feeder (train.py:45), eval.py synthesises a chosen one."""
    lj = str(tmp_path / "lj")
    os.makedirs(lj)
    _corpus(lj)
    vctk = str(tmp_path / "vctk")
    rng = np.random.default_rng(1)
    for spk, f0 in (("225", 120.0), ("301", 210.0)):
        os.makedirs(os.path.join(vctk, "wav48", "p" + spk))
        os.makedirs(os.path.join(vctk, "txt", "p" + spk))
        for i in range(2):
            L = int(20000 * rng.uniform(0.5, 0.8))
            t = np.arange(L) / 20000.0
            y = 0.5 * np.sin(2 * np.pi * f0 * t) + rng.normal(0, 0.01, L)
            name = "p%s_%03d" % (spk, i + 1)
            with wave.open(os.path.join(vctk, "wav48", "p" + spk, name + ".wav"), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(20000)
                f.writeframes((np.clip(y, -1, 1) * 32767).astype("<i2").tobytes())
            with open(os.path.join(vctk, "txt", "p" + spk, name + ".txt"), "w") as f:
                f.write("Please call Stella.\n")
    logs = str(tmp_path / "logs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--ljspeech", lj, "--vctk", vctk, "--model", "taco2",
                        "--log_dir", logs, "--hparams", SMALL, "--checkpoint_interval", "2", "--max_steps", "2",
                        "--precision", "bf16"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Loaded 3 different speaker(s)" in r.stdout
    run = os.path.join(logs, "logs-taco2")
    ck = os.path.join(run, "model.ckpt-2")
    assert os.path.exists(ck)
    import torch
    sd = torch.load(ck, map_location="cpu")
    assert tuple(sd["model/inference/speaker/speaker_embed"].shape) == (3, 16)
    assert tuple(sd["model/inference/decoder/attention_lstm/kernel"].shape) == (128 + 128 + 64, 4 * 64)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "eval.py"), "--checkpoint", ck, "--model", "taco2",
                        "--hparams", SMALL + ",num_speakers=3", "--precision", "bf16", "--speaker", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(os.path.join(run, "eval-2-0.wav"))
    # a single-speaker hparams set cannot load this checkpoint: refused, not silently truncated
    r = subprocess.run([sys.executable, os.path.join(ROOT, "eval.py"), "--checkpoint", ck, "--model", "taco2",
                        "--hparams", SMALL, "--precision", "bf16"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


def test_wavenet_train_then_generate(tmp_path):
    """train_wavenet.py (train_wavenet.py:19-135) on the same tiny corpus with a narrow network, then
    generate_wavenet.py (generate_wavenet.py:48-171) from its checkpoint: the persistent incremental generator and the
    full-window path, which carries the reference's own temperature-1.0 consistency check (:133-138)."""
    data = str(tmp_path / "lj")
    os.makedirs(data)
    _corpus(data)
    logs = str(tmp_path / "logs")
    small = "dilations_length=4,dilations_depth=2,skip_channels=64,sample_size=400,batch_size=4,queue_size=16"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_wavenet.py"), "--ljspeech", data, "--log-dir", logs,
                        "--model", "simple_wavenet", "--hparams", small, "--max-steps", "3", "--checkpoint-interval", "3",
                        "--summary-interval", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    run = os.path.join(logs, "simple_wavenet")
    ckpt = os.path.join(run, "model.ckpt-3")
    assert os.path.exists(ckpt) and "Step 3 " in r.stdout and os.path.exists(os.path.join(run, "events.jsonl"))
    for fast, n in (("true", "300"), ("false", "12")):
        out = str(tmp_path / ("gen_%s.wav" % fast))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "generate_wavenet.py"), ckpt, "--samples", n, "--hparams", small,
                            "--fast_generation", fast, "--wav_out_path", out, "--wav_seed", os.path.join(data, "wavs", "utt0.wav")],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        with wave.open(out, "rb") as f:
            assert f.getframerate() == 16000 and f.getnframes() >= int(n)


def test_speaker_conditioned_wavenet_train_then_generate(tmp_path):
    """--model wavenet (the reference's default, train_wavenet.py:108) with gc_channels: every piece conditioned on its
    speaker id from the feeder (:40-49), biases on; generate_wavenet.py --gc_channels / --gc_cardinality / --gc_id
    (:214-231) draws one speaker's voice through the incremental generator and through the full-window path."""
    lj = str(tmp_path / "lj")
    os.makedirs(lj)
    _corpus(lj)
    logs = str(tmp_path / "logs")
    small = ("dilations_length=4,dilations_depth=2,skip_channels=64,sample_size=400,batch_size=4,queue_size=16,"
             "gc_channels=8,use_biases=true")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_wavenet.py"), "--ljspeech", lj, "--log-dir", logs,
                        "--hparams", small, "--max-steps", "3", "--checkpoint-interval", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ckpt = os.path.join(logs, "wavenet", "model.ckpt-3")
    assert os.path.exists(ckpt) and "Loaded 1 different speaker(s)" in r.stdout
    for fast, n in (("true", "200"), ("false", "8")):
        out = str(tmp_path / ("gen_%s.wav" % fast))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "generate_wavenet.py"), ckpt, "--samples", n, "--hparams", small,
                            "--gc_channels", "8", "--gc_cardinality", "1", "--gc_id", "0", "--fast_generation", fast,
                            "--wav_out_path", out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        with wave.open(out, "rb") as f:
            assert f.getnframes() >= int(n)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_wavenet.py"), "--ljspeech", lj, "--log-dir", logs,
                        "--hparams", small + ",lc_channels=4", "--max-steps", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "lc_channels" in (r.stdout + r.stderr)
