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
            "--hparams", SMALL, "--checkpoint_interval", "2", "--precision", "bf16"]
    r = subprocess.run(base + ["--max_steps", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    run = os.path.join(logs, "logs-taco2")
    assert os.path.exists(os.path.join(run, "model.ckpt-2")) and os.path.exists(os.path.join(run, "train.log"))
    assert os.path.exists(os.path.join(run, "step-000002-audio.wav"))
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
