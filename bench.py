#!/usr/bin/env python3
"""Headline benchmark: Tacotron-2 training step, LJSpeech shapes (batch 32/GPU, T_in 160,
T_out 1000 mel frames, r=5), bf16 MFMA operands, synthetic data, data-parallel over N GPUs.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (Measurement)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nspeech_amd import hparams as hparams_mod  # noqa: E402
from nspeech_amd.models import create_model  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def synthetic_speech(rng, seconds, sr):
    """5 harmonics of f0 ~ U[90, 250] Hz with 4 Hz amplitude modulation + N(0, 0.01) noise, peak 0.8."""
    L = int(seconds * sr)
    t = np.arange(L) / sr
    f0 = rng.uniform(90, 250)
    y = sum(np.sin(2 * np.pi * f0 * (h + 1) * t) / (h + 1) for h in range(5)) * (0.5 + 0.5 * np.sin(2 * np.pi * 4 * t))
    return (0.8 * y / np.abs(y).max() + rng.normal(0, 0.01, L)).astype(np.float32)


def synthetic_batch(hp, N, Ti, To, seed, features=None):
    """SURVEY 8d inputs: ids ~ U{2..63} with lengths ~ U{T_in/2..T_in}, EOS last; targets = the reference-style
    feature extraction (A3 / A4) of synthetic speech-like audio.  `features(wav) -> (linear [F,T], mel [M,T])`: the GPU
    kernel in the benchmark, the NumPy oracle in the CPU baseline."""
    rng = np.random.default_rng(seed)
    lengths = rng.integers(Ti // 2, Ti + 1, size=N).astype(np.int32)
    inputs = np.zeros((N, Ti), np.int32)
    for n in range(N):
        inputs[n, :lengths[n] - 1] = rng.integers(2, 64, size=lengths[n] - 1)
        inputs[n, lengths[n] - 1] = 1
    if features is None:
        from nspeech_amd.utils import audio as A
        features = A.spectrogram_and_mel
    hop = int(hp.frame_shift_ms / 1000 * hp.sample_rate)
    mel = np.zeros((N, To, hp.num_mels), np.float32)
    lin = np.zeros((N, To, hp.num_freq), np.float32)
    for n in range(N):
        l, m = features(synthetic_speech(rng, To * hop / hp.sample_rate, hp.sample_rate))     # [F, 1 + To], [M, 1 + To]
        lin[n], mel[n] = l.T[:To], m.T[:To]
    return inputs, lengths, mel, lin


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(hp, seed):
    """The CPU restatement of the reference (oracle/, torch-CPU fp32) timed on a bounded sample of the same workload:
    whole training steps (forward, backward, clip_by_global_norm, Adam) at batch 8, T_in 160, T_out 500."""
    from oracle import audio_oracle as AO
    from oracle import taco2_oracle as O
    from nspeech_amd.models import params as P
    from nspeech_amd.utils.text.symbols import symbols
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))      # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(cores)
    sys.stderr.write("[bench] cpu baseline on %d threads...\n" % cores)
    sys.stderr.flush()
    lay, st = P.taco2_layout(hp, len(symbols))
    pv, sv = P.init_values(lay, st, seed)
    N, Ti, To = 8, 160, 500
    hpd = hp.values()
    inputs, lengths, mel, lin = synthetic_batch(hp, N, Ti, To, seed,
                                                features=lambda w: (AO.spectrogram(w, hpd), AO.melspectrogram(w, hpd)))
    p = {k: torch.tensor(v, requires_grad=True) for k, v in pv.items()}
    stats = {k: torch.tensor(v) for k, v in sv.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    vv = {k: torch.zeros_like(v) for k, v in p.items()}
    ti, tl, tm, tn = torch.tensor(inputs), torch.tensor(lengths), torch.tensor(mel), torch.tensor(lin)
    t0 = time.time()
    steps = 0
    while steps < 16 and (steps == 0 or time.time() - t0 < 12.0):    # about 10-30 s of CPU work
        for v in p.values():
            v.grad = None
        out = O.taco2_forward(dict(p, **stats), hpd, ti, tl, tm, tn)
        loss, _, _ = O.taco2_loss(hpd, out, tm, tn)
        loss.backward()
        with torch.no_grad():
            g, _ = O.clip_by_global_norm({k: v.grad for k, v in p.items()}, 1.0)
            q = {k: v.detach() for k, v in p.items()}
            O.adam_step(q, g, m, vv, steps + 1, O.learning_rate(hpd, steps), hp.adam["beta1"], hp.adam["beta2"])
            for k in p:
                p[k].copy_(q[k])
            for k, v in out["bn_updates"].items():
                stats[k] = v
        steps += 1
    dt = time.time() - t0
    return {"value": steps * N * To / dt, "unit": "mel-frames/s", "cores": cores, "cpu_model": cpu_model_name(),
            "kind": "port",
            "sample": "%d whole train steps (fwd + bwd + clip + Adam) at batch %d, T_in %d, T_out %d, fp32 torch-CPU "
                      "restatement of the reference (oracle/), %.1f s" % (steps, N, Ti, To, dt)}


def griffin_lim_bench(hp, with_cpu):
    """Second headline metric: Griffin-Lim real-time factor on 10 s of audio (60 iterations)."""
    from nspeech_amd.utils import audio as A
    from oracle import audio_oracle as AO
    hpd = hp.values()
    rng = np.random.default_rng(1234)
    L = 200000
    t = np.arange(L) / hp.sample_rate
    f0 = rng.uniform(90, 250)
    y = sum(np.sin(2 * np.pi * f0 * (h + 1) * t) / (h + 1) for h in range(5)) * (0.5 + 0.5 * np.sin(2 * np.pi * 4 * t))
    y = (0.8 * y / np.abs(y).max() + rng.normal(0, 0.01, L)).astype(np.float32)
    # normalised linear spectrogram with real dynamics (the shipped +100 min_level_db saturates)
    spec = AO.spectrogram(y, dict(hpd, min_level_db=-100)).T[:797].copy()
    st = torch.tensor(spec, device="cuda")
    for _ in range(2):
        A.griffin_lim_gpu(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        wav = A.griffin_lim_gpu(st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    audio_s = wav.numel() / hp.sample_rate
    # algorithmic HBM bytes per call: per iteration read S + read y + write y, plus the init pass
    T, F, win = 797, hp.num_freq, 1000
    alg = 60 * (T * F * 4 + 2 * wav.numel() * 4) + (T * F * 4 + wav.numel() * 4)
    res = {"rtf": ms * 1e-3 / audio_s, "ms": ms, "audio_s": audio_s, "iters": int(hp.griffin_lim_iters),
           "algorithmic_GBps": alg / (ms * 1e-3) / 1e9, "hbm_peak_GBps": HBM_PEAK_GBS}
    # batched as in the reference's training graph (tacotron.py:107): 32 clips in one call
    sb = st.unsqueeze(0).repeat(32, 1, 1).contiguous()
    A.griffin_lim_gpu(sb)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        A.griffin_lim_gpu(sb)
    e1.record()
    torch.cuda.synchronize()
    msb = e0.elapsed_time(e1) / 3
    res["batch_32"] = {"ms": msb, "rtf": msb * 1e-3 / (32 * audio_s), "algorithmic_GBps": 32 * alg / (msb * 1e-3) / 1e9}
    if with_cpu:
        sys.stderr.write("[bench] griffin-lim cpu sample...\n")
        sys.stderr.flush()
        t0 = time.time()
        AO.inv_spectrogram_tensorflow(spec, hpd)             # the same 797 frames, 60 iterations
        dt = time.time() - t0
        res["cpu_rtf"] = dt / ((796 * 250 + 1000) / hp.sample_rate)
        res["cpu_sample"] = "float64 NumPy oracle, the same 797 frames (10 s of audio), 60 iterations, 1 thread, %.1f s" % dt
    return res


def inference_bench(hp, dtype, seed=1234):
    """Free-running synthesis (SURVEY 8d, C2 eval / C4): mel frames per second at batch 1 and 32, max_iters 300
    (the reference default) - the whole pass: encoder, 300 decoder steps, postnet, expand net, linear head."""
    import copy
    out = {}
    for N in (1, 32):
        hpi = copy.deepcopy(hp)
        hpi.max_iters = 300
        m = create_model("taco2", hpi, device="cuda:%d" % torch.cuda.current_device(), dtype=dtype, seed=seed)
        inputs, lengths, _, _ = synthetic_batch(hpi, N, 160, 10, seed)
        for _ in range(3):                       # eager pass, graph capture, first replay
            m.initialize(inputs, lengths)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            m.initialize(inputs, lengths)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        frames = N * 300 * hpi.outputs_per_step
        out["batch_%d" % N] = {"mel_frames_per_s": frames / dt, "ms": dt * 1e3, "decoder_steps": 300,
                               "rtf": dt / (300 * hpi.outputs_per_step * hpi.frame_shift_ms * 1e-3),
                               # persistent: ns_taco2_decode (one launch); rows32: packed step products, 8 launches a step
                               "decode_path": m.last_paths.get("decode")}
        del m
    return out


def taco1_bench(args, features, seed=1234):
    """BASELINE config 1's model (Tacotron-1: CBHG encoder, Bahdanau attention with a GRU cell, residual GRU decoder, post
    CBHG; hparams/taco1.yaml widths) at the benchmark's shape: one training step = forward + losses + backward + clip + Adam
    on a device-resident synthetic batch of 32 x T_in 160 x T_out 1000 (r = 5), the same precision mode as the headline.
    `gru`: the four persistent GRU recurrences (ns_gru_seq_*, csrc/gru.hip) timed alone with HIP events on the launch
    stream; `roofline` = their gate products (recurrent halves in the loop + hoisted input halves are NOT in these
    launches: recurrent flop only) against the bf16 MFMA peak - like the LSTM recurrences of the headline they sit at
    their step latency, not at any throughput bound."""
    from nspeech_amd import hparams as hparams_mod, ops
    hp1 = hparams_mod.load("taco1")
    N, Ti, To = args.batch, args.t_in, args.t_out
    m = create_model("taco1", hp1, device="cuda:%d" % torch.cuda.current_device(), dtype=args.dtype, seed=seed)
    m.add_optimizer(global_step=0)
    inputs, lengths, mel, lin = synthetic_batch(hp1, N, Ti, To, seed, features=features)
    m.initialize(inputs, lengths, None, mel, lin)              # uploads the batch once; the timed steps re-run it from HBM
    step = lambda: m.step(read_loss=False)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    m.check_status()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    m.check_status()
    # the GRU launches alone: events around every ns_gru_seq call of one more step
    ev, orig = [], ops.gru_seq

    def timed(direction, p0, p1, work):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(direction, p0, p1, work)
        e1.record()
        ev.append((direction, p0.H, p0.T, 2 if p1 is not None else 1, e0, e1))
    ops.gru_seq = timed
    try:
        step()
        torch.cuda.synchronize()
    finally:
        ops.gru_seq = orig
    gru, flop, ms_sum = [], 0.0, 0.0
    for direction, H, T, nd, e0, e1 in ev:
        ms = e0.elapsed_time(e1)
        f = nd * T * 2.0 * N * H * 3 * H                        # recurrent gate products of the launch
        gru.append({"pass": direction, "H": H, "steps": T, "directions": nd, "ms": ms, "us_per_step": ms * 1e3 / T,
                    "TFLOPs": f / (ms * 1e-3) / 1e12})
        flop += f
        ms_sum += ms
    out = {"ms_per_step": dt * 1e3, "mel_frames_per_s": N * To / dt, "precision_mode": args.dtype,
           "config": {"workload": "Tacotron-1 train step (fwd+bwd+clip+Adam), batch %d, T_in %d, T_out %d, r=%d, taco1.yaml widths"
                                  % (N, Ti, To, hp1.outputs_per_step)},
           "paths": dict(m.last_paths), "gru": gru,
           "roofline": {"bound": "mfma", "kernel": "gru_fwd_kernel / gru_bwd_kernel (persistent GRU recurrences, %d launches)" % len(ev),
                        "achieved": flop / (ms_sum * 1e-3) / 1e12 if ms_sum else None, "peak": MFMA_BF16_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": flop / (ms_sum * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS if ms_sum else None,
                        "ms": ms_sum, "binding_bound": "step latency of a two-product recurrence (LDS barriers at H = 128, two "
                                                       "CU-to-CU hops at H = 256), not MFMA throughput"}}
    del m
    return out


def wavenet_bench(seed=1234):
    """BASELINE config 4 (simple_wavenet, shipped wavenet.yaml: 50 layers, receptive field 5117): one training step on
    8 clips of receptive field + 8000 samples, and incremental generation of 2000 samples behind a receptive-field
    seed, batch 1 and 32 (one workgroup per waveform), bf16 weights."""
    from nspeech_amd.models.wavenet import mu_law_encode, receptive_field
    hp = hparams_mod.load("wavenet")
    rf = receptive_field(hp)
    m = create_model("simple_wavenet", hp, device="cuda:%d" % torch.cuda.current_device(), dtype="bf16", seed=seed)
    m.add_optimizer(0)
    rng = np.random.default_rng(seed)
    N, T = 8, rf + 8000
    t = np.arange(T) / 16000.0
    audio = (0.5 * np.sin(2 * np.pi * 220 * t)[None] + 0.02 * rng.standard_normal((N, T))).astype(np.float32)
    for _ in range(2):
        m.step(audio)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.step(audio)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    out = {"receptive_field": rf, "train": {"ms_per_step": dt * 1e3, "target_samples_per_s": N * (T - rf) / dt,
                                            "batch": N, "clip_samples": T, "loss": m.loss}}
    seeds = mu_law_encode((0.01 * rng.standard_normal((1, rf))).astype(np.float32), hp.quantization_channels)
    for B in (1, 32):
        sd = np.repeat(seeds, B, 0)
        m.generate(sd, 64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.generate(sd, 2000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        m.generate(sd, 1)
        torch.cuda.synchronize()
        warm = time.perf_counter() - t0                      # the seed walk alone
        gen = max(dt - warm, 1e-9)
        out["generate_batch_%d" % B] = {"samples_per_s": B * 2000 / gen, "us_per_drawn_sample": gen / 2000 * 1e6,
                                        "seed_walk_ms": warm * 1e3, "realtime_factor_16k": (B * 2000 / gen) / 16000.0,
                                        "engine": {1: "single-wave VALU chain", 2: "MFMA chain + skip waves",
                                                   3: "MFMA chain + skip waves + 4 post-processing helper workgroups per waveform"
                                                   }.get(getattr(m, "last_engine", 0), "per-layer kernel")}
    return out


def real_dynamics_bench(hp, model, args, one_step):
    """The timed step again on targets that are not saturated: the synthetic speech of SURVEY 8d through the feature
    kernels with min_level_db = -100 (the value the reference's comment intends).  Same shapes, same kernels;
    what can differ is data-dependent work - none in this path - and the loss.  10 steps after 3."""
    old = hp.min_level_db
    hp.min_level_db = -100
    try:
        inputs, lengths, mel, lin = synthetic_batch(hp, args.batch, args.t_in, args.t_out, 1234)
    finally:
        hp.min_level_db = old
    sat = float((np.abs(mel - 1.0) < 1e-6).mean())
    model.initialize(inputs, lengths, None, mel, lin)
    for _ in range(3):
        one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        one_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    return {"ms_per_step": dt * 1e3, "value": args.batch * args.t_out / dt, "loss": model.read_losses(),
            "mel_target_mean": float(mel.mean()), "mel_target_std": float(mel.std()), "mel_targets_saturated_frac": sat,
            "note": "targets = features of the same synthetic speech with min_level_db = -100 (not saturated)"}


def e2e_bench(hp, model, args, device, modes=("hbm_cache", "hbm_cache_sync_loss", "host_cache")):
    """train.py's own loop at the benchmark shape on synthetic FILES: an LJSpeech-layout corpus of 64
    utterances of 12.45 s (-> T_out 1000) and 80-159 characters is written to a temporary directory; the feeder thread
    reads the wavs, extracts the features on the GPU, sorts / pads / deals batches as the reference does, and
    train.train_step() runs the step and reads the loss back EVERY step (train.py:80).  Timed after one pass over the
    corpus (features cached as the reference caches them).  Two forms: features cached in HBM and batches assembled on
    the device (train.py's default), and features cached in host RAM with pinned uploads on a copy stream."""
    import copy
    import shutil
    import tempfile
    import train as train_cli
    from nspeech_amd.datasets.datafeeder import DataFeeder, DeviceStager
    from nspeech_amd.utils import audio as A
    rng = np.random.default_rng(4321)
    tmp = tempfile.mkdtemp(prefix="nspeech_e2e_")
    out = {}
    try:
        os.makedirs(os.path.join(tmp, "wavs"))
        hop = int(hp.frame_shift_ms / 1000 * hp.sample_rate)
        L = (args.t_out - 4) * hop          # T = 1 + L // hop = t_out - 3 frames -> padded to t_out (datafeeder.py:204-206)
        letters = np.array(list("abcdefghijklmnopqrstuvwxyz    "))
        with open(os.path.join(tmp, "metadata.csv"), "w") as f:
            for i in range(2 * args.batch):
                A.save_wav(synthetic_speech(rng, L / hp.sample_rate, hp.sample_rate)[:L], os.path.join(tmp, "wavs", "U%03d.wav" % i))
                n = int(rng.integers(args.t_in // 2, args.t_in))
                text = "a" + "".join(rng.choice(letters, n - 2)) + "z"
                f.write("U%03d|%s|%s\n" % (i, text, text))
        hpf = copy.deepcopy(hp)
        hpf.batch_size, hpf.batch_group_size = args.batch, 2
        for key, dev_cache, sync in (("hbm_cache", True, False), ("hbm_cache_sync_loss", True, True), ("host_cache", False, False)):
            if key not in modes:
                continue
            feeder = DataFeeder(hpf, ljspeech=tmp, seed=7, pinned=True, device_cache=dev_cache).start()
            batches = DeviceStager(feeder, device)
            pipe = None if sync else train_cli.LossPipeline(model, batches)
            run = (lambda: train_cli.train_step(model, batches)) if sync else pipe.step
            for _ in range(4):                          # first pass over the corpus: wav reads + feature extraction
                run()
            if pipe is not None:
                pipe.drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            last = None
            for _ in range(args.e2e_steps):
                last = run()
            if pipe is not None:
                last = pipe.drain()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.e2e_steps
            loss = last if sync else last[1]
            To = int(model.mel_targets.shape[1])
            out[key] = {"e2e_ms_per_step": dt * 1e3, "mel_frames_per_s": args.batch * To / dt, "t_out": To,
                        "t_in": int(model.inputs.shape[1]), "loss": loss}
            feeder.stop()                               # (its thread, queued batches and pinned memory would outlive this block)
            del feeder, batches, pipe, run
        out["note"] = ("train.py's loop (feeder thread -> batch on the GPU -> step -> EVERY step's loss read back) on %d "
                       "synthetic wav files; hbm_cache: features stay in HBM, batches assembled on the device, the loss "
                       "read-back one step behind the launches (train.py's defaults); hbm_cache_sync_loss: the same with "
                       "the host waiting for each step's loss before it issues the next (--sync-loss); host_cache: "
                       "features in RAM, pinned H2D on a copy stream (--feature-cache host)" % (2 * args.batch))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def launch_ranks(n):
    """Start `n` ranks of this script under torch.distributed.run on this node and relay their output.  The driver's own
    multi-GPU runs launch torch.distributed.run themselves; this is the same command line."""
    import socket
    import subprocess
    with socket.socket() as sk:                 # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] starting %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # 50 steps after 10 warm-up (2.2 s timed)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--t-in", type=int, default=160)
    ap.add_argument("--t-out", type=int, default=1000)
    ap.add_argument("--dtype", default="mixed", choices=["mixed", "bf16", "bf16x3", "fp32"],
                    help="mixed (default): split-bf16 x3 on the mel path forward, bf16 elsewhere - meets the 1e-3 mel tolerance")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phases", action="store_true", help="print a per-phase time table to stderr")
    ap.add_argument("--train-only", action="store_true",
                    help="skip the Griffin-Lim / synthesis / WaveNet legs (profiling passes of the headline workload)")
    ap.add_argument("--e2e-steps", type=int, default=12,
                    help="steps of the end-to-end leg (feeder thread + pinned H2D + per-step loss read-back); 0 = skip")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as a CHILD process (never re-exec a process that may
        # have touched the GPU) and relay rank 0's JSON line.  Nothing above has made a HIP call.
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or let bench.py start "
                         "the ranks itself: run it without torch.distributed.run)\n" % (args.gpus, world, args.gpus))
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # NSPEECH_DIST_BACKEND=gloo lets the multi-rank flow be rehearsed with several ranks on ONE GPU (tests); the
    # driver's runs use nccl = RCCL, one rank per GPU
    backend = os.environ.get("NSPEECH_DIST_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    hp = hparams_mod.load("taco2")
    model = create_model("taco2", hp, device="cuda:%d" % local, dtype=args.dtype, seed=1234, world_size=world)
    if world > 1:
        from nspeech_amd import parallel
        parallel.broadcast_parameters(model, 0)
        model.reducer = parallel.make_reducer(model)
    inputs, lengths, mel, lin = synthetic_batch(hp, args.batch, args.t_in, args.t_out, 1234 + rank)
    model.add_optimizer(global_step=0)
    model.initialize(inputs, lengths, None, mel, lin)

    def one_step():
        model.forward_train()
        model.backward()          # hands each gradient bucket to RCCL as soon as it is final
        model.apply_gradients()   # waits for the reductions, then clip + Adam on the summed gradients

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        one_step()
        marks[i + 1].record()        # an event record does not synchronise: per-step times for the median
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = model.read_losses()
    # who took part: every rank reports its device, rank 0 checks the count against --gpus
    seen, devices = world, [torch.cuda.get_device_name(local)]
    if world > 1:
        seen = dist.get_world_size()
        names = [None] * world
        dist.all_gather_object(names, "rank %d: cuda:%d %s" % (rank, local, torch.cuda.get_device_name(local)))
        devices = names
        if seen != args.gpus:
            raise RuntimeError("bench.py: --gpus %d but the process group has %d ranks" % (args.gpus, seen))
    ms = dt / args.steps * 1e3
    frames = args.batch * args.t_out * world

    # Two more steps outside the timed region, executed by EVERY rank (the step contains collectives):
    # one with per-phase events, one with an event pair around every large GEMM launch.
    # The phase pass runs every launch on ONE stream (the timed steps put the weight gradients on a second one): a
    # phase then holds all of its own work - the decoder-gate roofline below counts the weight-gradient products of
    # the decoder LSTMs - and the phases add up to the single-stream step, a little more than ms_per_step.
    overlap, model.overlap_wgrads = getattr(model, "overlap_wgrads", False), False
    one_step()                      # untimed: the one-stream form names some buffers differently (first-touch fills)
    torch.cuda.synchronize()
    model.timing = []
    one_step()
    torch.cuda.synchronize()
    tm = model.timing
    model.timing = None
    model.overlap_wgrads = overlap
    phases = [(tm[i][0], tm[i - 1][1].elapsed_time(tm[i][1])) for i in range(1, len(tm))]
    if args.phases and rank == 0:
        for name, v in phases:
            sys.stderr.write("%-20s %8.3f ms\n" % (name, v))
        sys.stderr.write("%-20s %8.3f ms\n" % ("total", sum(v for _, v in phases)))
    from nspeech_amd import profiling
    roof = profiling.roofline(model, one_step)

    if rank == 0:
        # sub-phases ("dec_lstm:loop1") fold into their phase for the table; the parts feed the gate-GEMM roofline
        pd = {}
        for name, v in phases:
            pd[name.split(":")[0]] = pd.get(name.split(":")[0], 0.0) + v
        sub = dict(phases)
        # decoder gate GEMMs (the figure north_star asks for): 2*N*[(768+1024)+(1024+1024)]*4096 flop per
        # decoder step forward, x3 with the backward, over the time of the two decoder-LSTM phases (hoisted input
        # GEMMs + the recurrent step launches + their weight-gradient GEMMs), HIP events on the launch stream
        S = args.t_out // hp.outputs_per_step
        D = hp.decoder_lstm_units
        gate_flop = 3 * 2.0 * args.batch * ((hp.attention_dim + 2 * hp.encoder_lstm_units + D) + (D + D)) * 4 * D * S
        gate_ms = pd.get("dec_lstm", 0.0) + pd.get("dec_lstm_bwd", 0.0)
        step_flop = 2.0 * args.batch * D * 4 * D              # one recurrent launch: [N, D] x [D, 4D]
        loops = {k: sub.get(k, 0.0) for k in ("dec_lstm:loop1", "dec_lstm:loop2", "dec_lstm_bwd:loop2", "dec_lstm_bwd:loop1")}
        from nspeech_amd import profiling as _prof
        state_mb = args.batch * D * 4 / 1e6
        gate_roof = {
            "bound": "mfma", "kernel": "decoder LSTM(1024) x 2 gate GEMMs: lstm_wide_fwd_kernel / lstm_wide_bwd_ps_kernel (ONE "
                                       "persistent launch per LSTM and direction, W_h register-resident, %d steps each) + the "
                                       "hoisted input and weight-gradient GEMMs" % S,
            "achieved": gate_flop / (gate_ms * 1e-3) / 1e12 if gate_ms else None, "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": gate_flop / (gate_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS if gate_ms else None,
            "gflop_fwd_bwd": gate_flop / 1e9, "ms": gate_ms,
            "recurrence": {k.replace("dec_lstm", "").strip(":_"): {"launches": 1, "steps": S, "us_per_step": v * 1e3 / S,
                                                                   "TFLOPs": step_flop / (v / S * 1e-3) / 1e12 if v else None}
                           for k, v in loops.items()},
            "binding_bound": "weight-stationary: W_h never leaves the registers, so the HBM weight stream of the launch-per-step "
                             "form is gone; each step is one store -> visible -> load hop of the state between the CUs "
                             "(forward: %.2f MB of state published and gathered by every workgroup per step; backward: "
                             "partial dh sums as self-flagging granules, one hop); this run's per-step times are in "
                             "`recurrence`, the split of a step into hop and arithmetic is traced in "
                             "profiles/r03_wide_trace.txt" % state_mb,
            "traffic": _prof.pmc_traffic("lstm_wide"),
            "traffic_note": "bytes per LAUNCH (= %d steps) at the L2's fabric side, %s" % (S, _prof.pmc_traffic_source()),
        }
        res = {
            "metric": "mel-frames/sec Tacotron-2 LJSpeech bs32 train step", "value": frames / (dt / args.steps),
            "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "ms_per_step_median": median_ms, "value_at_median": args.batch * args.t_out * world / (median_ms * 1e-3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"mixed": "bf16", "bf16": "bf16", "bf16x3": "bf16", "fp32": "f32"}[args.dtype], "precision_mode": args.dtype,
            "data": "synthetic",
            "config": {"workload": "Tacotron-2 train step (fwd+bwd+clip+Adam), batch %d/GPU, T_in %d, T_out %d, r=%d; "
                                   "targets = spectrogram features of synthetic speech (SURVEY 8d)"
                                   % (args.batch, args.t_in, args.t_out, hp.outputs_per_step),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "loss": loss},
            "roofline": gate_roof,
            "roofline_gemm_family": roof,
            "phases_ms": {k: round(v, 3) for k, v in pd.items()},
            "phases_note": "one extra step with every launch on one stream (sum = single-stream step); the timed steps "
                           "run the weight-gradient products on a second stream",
        }
        res["ranks_seen"] = seen
        res["rank_devices"] = devices
        if world == 1 and not args.train_only:
            # the same step on targets with real dynamics: under the shipped min_level_db = +100 the
            # normalisation saturates and every target frame is the same vector; with -100 the features of the
            # same synthetic speech spread over [0, 1]
            res["real_dynamics"] = real_dynamics_bench(hp, model, args, one_step)
            # the same step with hparams.deterministic_gradients (every sum of the backward pass in a fixed order)
            model.deterministic = True
            for _ in range(3):
                one_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                one_step()
            torch.cuda.synchronize()
            dtd = (time.perf_counter() - t0) / 10
            model.deterministic = False
            res["deterministic_gradients"] = {"ms_per_step": dtd * 1e3, "value": args.batch * args.t_out / dtd,
                                              "note": "bit-reproducible gradients (split-K partial tiles parked and added in "
                                                      "slice order); the headline runs the default, fp32-atomic split-K"}
            if args.e2e_steps > 0:
                em = create_model("taco2", hp, device="cuda:%d" % local, dtype=args.dtype, seed=1234)
                em.add_optimizer(global_step=0)
                res["e2e"] = e2e_bench(hp, em, args, "cuda:%d" % local)
                del em
            res["griffin_lim"] = griffin_lim_bench(hp, not args.no_cpu_baseline)
            res["inference"] = inference_bench(hp, args.dtype)
            res["wavenet"] = wavenet_bench()
            res["taco1"] = taco1_bench(args, None)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(hp, 1234)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
