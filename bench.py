#!/usr/bin/env python3
"""Headline benchmark: 1 s / 16 kHz clips per second, PCM in HBM -> logits in HBM (mel + CNN + LSTM).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 4096] [--arch simple]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
one rank per GPU; the batch is sharded by clip (4096 clips PER GPU, weak scaling), every rank runs
K1 (log-mel) -> K2 (conv stack + pool) -> K3 (LSTM + fc) on its shard and the per-clip logits are
all-gathered over RCCL so that every rank holds all N x 4096 x 2 logits.

A "step" = one pass of the whole path over one batch that is already resident in HBM.  K steps are
timed between barrier + torch.cuda.synchronize() on both sides; the slowest rank's time is used.
Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline      the dominant kernel (K2, f32 MFMA): algorithmic flops / its average launch duration
                (HIP events on the launch stream inside the timed region) / the dense f32 MFMA peak
  stages        the same for K1 (HBM roofline, 74,240 algorithmic bytes per clip) and K3 (gate GEMMs)
  cpu_baseline  the CPU oracle (numpy log-mel per clip + torch CPU Conv2d/LSTM/Linear) timed on this
                host on a bounded sample of the same clips (rank 0, N = 1 only)
  sustained     >= 2 s of back-to-back steps after the timed region (the driver's K steps last ~20 ms: too short to show the
                clock the chip holds under load): clips/s and per-kernel means over that leg; `timed_region_again` = the same K steps
                timed once more right behind it (the K-step figure at the settled clock; `value` stays the cold run's)
  parity        max |err| of the measured path against that oracle on the sample
  augmentation  augment_audio on the GPU: clips/s for plans drawn like the reference, error vs the oracle
  training      SURVEY 8(f).3: training step of SimpleWakewordModel at the same batch (train-mode forward + backward + Adam), both
                arithmetics; training_pipeline: augment -> log-mel -> step; training_3conv: the notebook's 3-conv model
  streaming     BASELINE configs[4] (256 microphones, 10 ms hop, hipGraph replay per hop): p50/p99 hop latency, hops/s
                (rank 0, N = 1 only; measured after the timed region)
  forward_3conv the notebook's 3-conv WakewordModel at the same batch: clips/s, per-stage ms, roofline of its conv stack
  decode        K0 alone (sample bytes in HBM -> normalised clips), 16 kHz and 48 kHz
  file_pipeline WAV files in a temporary directory -> logits (reader threads -> pinned staging -> H2D -> K0..K3): the file-fed rate,
                to be read beside the reference's own published 453 clips/s (wakeword_training.ipynb:742)
  gather        when a process group exists (N > 1, or WW_BENCH_FORCE_DIST=1 for a one-rank RCCL group): blocking latency of the
                logits all-gather alone
Environment: WW_BENCH_BACKEND=gloo is a REHEARSAL mode (ranks share GPUs, host-staged gather) and is labelled as such in the line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic work per clip, SURVEY.md section 8(d) (T = 32 frames)
K1_BYTES_PER_CLIP = 16000 * 4 + 80 * 32 * 4            # 74,240: PCM read once + log-mel written once
K1_FLOPS_PER_CLIP = 2.1e6                              # 32 x 2.5 N log2 N real FFTs + window/power/sparse mel/log
K2_FLOPS_PER_CLIP = {"simple": 1_474_560 + 94_371_840, "full": 1_474_560 + 94_371_840 + 377_487_360}
K3_FLOPS_PER_CLIP = {"simple": 98_304 + 393_216 + 1_024, "full": 196_608 + 393_216 + 1_024}   # live gates i,g,o only
HBM_PEAK = 8.0e12                                      # B/s, MI355X_MICROARCH.md chip table (spec)
MFMA_F32_PEAK = 157.3e12                               # flop/s, dense f32 MFMA = f32 vector peak (same table)
MFMA_F16_PEAK = 2.5e15                                 # flop/s, dense f16/bf16 MFMA (same table, "~2.5 PF dense")
METRIC = "1s/16kHz clips/sec end-to-end (mel+CNN+LSTM)"


def pmc_traffic(kernel, batch, arch, with_source=False):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r*_pmc_traffic.json: FETCH_SIZE x2 +
    WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md).  Counters cannot be read from inside this process;
    the newest committed pass for this batch/arch is quoted (its file name goes into `traffic_source`), else null."""
    if with_source:
        for path in _pmc_files(arch):
            v = _pmc_traffic_from(path, kernel, batch)
            if v is not None:
                return v, "profiles/" + os.path.basename(path)
        return None, None
    import glob
    if batch != 4096:
        return None
    pattern = "r*_pmc_traffic.json" if arch == "simple" else "r*_full_pmc_traffic.json"      # the 3-conv passes are kept in files of their own
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        if arch == "simple" and "_full_" in path:
            continue
        try:
            k = json.load(open(path))["kernels"]
            hits = [d["hbm_bytes_per_launch_corrected"] for name, d in k.items() if any(part in name for part in kernel.split("+"))]
            if hits:
                return sum(hits)                     # "a+b": the stage is two launches (conv2 kernel + conv3 kernel)
        except Exception:
            continue
    return None


def _pmc_files(arch):
    import glob
    pattern = "r*_pmc_traffic.json" if arch == "simple" else "r*_full_pmc_traffic.json"
    return [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True) if arch != "simple" or "_full_" not in f]


def _pmc_traffic_from(path, kernel, batch):
    if batch != 4096:
        return None
    try:
        k = json.load(open(path))["kernels"]
        hits = [d["hbm_bytes_per_launch_corrected"] for name, d in k.items() if any(part in name for part in kernel.split("+"))]
        return sum(hits) if hits else None
    except Exception:
        return None


def pmc_busy(kernel, batch, arch):
    """Matrix-pipe busy fraction of a kernel from the newest committed PMC pass (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs x
    kernel cycles), profiles/r*_pmc_traffic.json) -- what north_star calls MFMA utilisation of the gate GEMM."""
    import glob
    if batch != 4096 or arch != "simple":
        return {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            for name, d in json.load(open(path))["sq_counters_mean"].items():
                if kernel in name and "mfma_pipe_busy_frac" in d:
                    return {"mfma_pipe_busy_frac_pmc": d["mfma_pipe_busy_frac"], "pmc_source": os.path.basename(path)}
        except Exception:
            continue
    return {}


def cpu_baseline(clips, sd, budget_s=20.0):
    """The reference path restated on the CPU (oracle/): per-clip numpy log-mel, then torch CPU layers."""
    from oracle import mel_oracle, model_oracle
    basis = mel_oracle.mel_filterbank()
    module = model_oracle.torch_module_from_state_dict(sd)
    mel_oracle.process_clip(clips[0], True, basis)                       # warm caches / FFT plans
    t0 = time.perf_counter()
    n, mels = 0, []
    while n < len(clips) and (time.perf_counter() - t0 < budget_s * 0.6 or n < 16):
        mels.append(mel_oracle.process_clip(clips[n], True, basis))
        n += 1
    t_mel = time.perf_counter() - t0
    x = torch.from_numpy(np.stack(mels)[:, None].astype(np.float32))
    with torch.no_grad():
        module(x[:2])
        t1 = time.perf_counter()
        logits = np.concatenate([module(x[s:s + 64]).numpy() for s in range(0, n, 64)])   # config-1 sized batches
        t_model = time.perf_counter() - t1
    total = t_mel + t_model
    return {
        "value": n / total, "unit": "clips/s", "cores": int(torch.get_num_threads()), "kind": "port",
        "sample": f"{n} of the bench clips, one at a time: numpy/scipy float64-FFT log-mel (1 thread) then torch CPU "
                  f"Conv2d/LSTM/Linear forward in batches of 64 ({torch.get_num_threads()} threads)",
        "mel_clips_per_s": n / t_mel, "model_clips_per_s": n / t_model, "host_cpus": os.cpu_count(),
    }, np.stack(mels)[:, None].astype(np.float32), logits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="clips per GPU per step")
    ap.add_argument("--arch", default="simple", choices=["simple", "full"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streaming", action="store_true", help="skip the streaming (256 mics, 10 ms hop) latency leg")
    ap.add_argument("--conv-math", default=None, choices=["f32", "f16x3", "f16x3d"], help="conv arithmetic (default: library default)")
    ap.add_argument("--logmel-math", default=None, choices=["f32", "f64", "auto"], help="log-mel arithmetic (default: library default, auto)")
    ap.add_argument("--sustained-s", type=float, default=2.0, help="length of the sustained leg after the timed region (0 = skip)")
    args = ap.parse_args()

    import wakeword_jupyterlab_amd as pkg
    from wakeword_jupyterlab_amd import distributed as wdist

    # WW_BENCH_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks (ranks share GPUs, the logits
    # all-gather goes through host memory); the driver's runs use RCCL ("nccl").
    backend = os.environ.get("WW_BENCH_BACKEND", "nccl")
    force_dist = os.environ.get("WW_BENCH_FORCE_DIST", "0") == "1"       # one rank, but with a real (RCCL) group around the step
    rank, world, local = wdist.init_from_env(backend, force=force_dist)
    if backend == "gloo" and torch.cuda.is_available():
        local = local % torch.cuda.device_count()
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only implementation of this path")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from wakeword_jupyterlab_amd import _native as nat
    from wakeword_jupyterlab_amd import ops

    ops.init()
    if args.conv_math:
        ops.set_conv_math(args.conv_math)
    conv_math = ops.get_conv_math()
    if args.logmel_math:
        ops.set_logmel_math(args.logmel_math)
    logmel_math = ops.get_logmel_math()
    B = args.batch
    n_conv = 2 if args.arch == "simple" else 3
    sd = pkg.synth.make_state_dict(args.arch, seed=1234)
    model = (pkg.SimpleWakewordModel() if args.arch == "simple" else pkg.WakewordModel())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    packed = model.packed_weights()

    # rank r holds clips [r*B, (r+1)*B) of the global batch: B DISTINCT synthetic clips (seed = clip index)
    host = pkg.synth.make_clips(rank * B, B)
    pcm = torch.from_numpy(host).to(dev)
    ws = torch.empty(nat.check(nat.lib.ww_workspace_bytes(B, n_conv)), device=dev, dtype=torch.uint8)
    mel = torch.empty((B, 1, 80, 32), device=dev)
    pooled = torch.empty((B, ops.c_last(n_conv)), device=dev)
    scratch_bytes = nat.check(nat.lib.ww_cnn_scratch_bytes(B, n_conv))
    scratch = torch.empty(max(1, scratch_bytes), device=dev, dtype=torch.uint8)
    # logits / gathered are double-buffered: the all-gather of step k runs on RCCL's stream while step k+1 computes
    # (distributed.LogitsGatherPipeline: the same object tests/_rccl_child.py drives on a one-rank RCCL group)
    pipe = wdist.LogitsGatherPipeline(B, dev)

    import ctypes as C
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    stream = torch.cuda.current_stream()
    st = C.c_void_p(stream.cuda_stream)

    def step(ev=None):
        logits = pipe.acquire()             # waits for the gather that last used this buffer pair (two steps ago)
        if ev: ev[0].record(stream)
        nat.check(nat.lib.ww_logmel_f32(p(pcm), B, 16000, 16000, 1, p(mel), st))
        if ev: ev[1].record(stream)
        nat.check(nat.lib.ww_cnn_pool_f32(p(mel), B, 32, p(packed), n_conv, p(scratch) if scratch_bytes else None, p(pooled), st))
        if ev: ev[2].record(stream)
        nat.check(nat.lib.ww_lstm_fc_f32(p(pooled), B, p(packed), n_conv, p(logits), st))
        if ev: ev[3].record(stream)
        pipe.submit()                       # async all-gather of this step's logits (nothing without a process group)

    def fence():
        pipe.drain()
        if pipe.active:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    fence()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], device=dev if backend != "gloo" else "cpu", dtype=torch.float64)
    if pipe.active:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # the exchange alone: blocking latency of one all-gather of [B, 2] logits (every rank takes part)
    gather_us = None
    if pipe.active and backend != "gloo":
        g_out = torch.empty((world * B, 2), device=dev)
        for _ in range(5):
            dist.all_gather_into_tensor(g_out, pipe.logits[0])
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for _ in range(50):
            dist.all_gather_into_tensor(g_out, pipe.logits[0])
        torch.cuda.synchronize()
        gather_us = 1e6 * (time.perf_counter() - tg) / 50

    ms = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(3)] for e in events])    # [steps, 3] K1,K2,K3
    k1_ms, k2_ms, k3_ms = [float(v) for v in ms.mean(axis=0)]

    # sustained leg: the same step back to back for >= sustained_s seconds (every rank runs it; rank 0 reports)
    sustained = None
    if args.sustained_s > 0:
        n_sus = max(50, int(args.sustained_s / max(1e-5, elapsed / args.steps)) + 1)
        ev_every = max(1, n_sus // 200)                       # events on ~200 of the steps
        sus_events = []
        fence()
        t1 = time.perf_counter()
        for k in range(n_sus):
            if k % ev_every == 0:
                e4 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                sus_events.append((k, e4))
                step(e4)
            else:
                step()
        fence()
        sus_elapsed = time.perf_counter() - t1
        half = [e for k, e in sus_events if k >= n_sus // 2]   # second half: the clock has settled
        sms = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(3)] for e in half])
        sustained = {"seconds": sus_elapsed, "steps": n_sus, "clips_per_s_per_gpu": B * n_sus / sus_elapsed,
                     "ms_per_step": 1e3 * sus_elapsed / n_sus,
                     "kernel_ms_second_half": dict(zip(["K1", "K2", "K3"], [float(v) for v in sms.mean(axis=0)])),
                     "note": "host-timed, barrier + synchronize on both sides, same step as the timed region"}
        # ... and the timed region once more, EXACTLY K steps, straight behind the sustained leg: the K-step figure at the settled clock
        # (reported beside `value`, which stays the cold run's)
        t2 = time.perf_counter()
        for k in range(args.steps):
            step()
        fence()
        again = time.perf_counter() - t2
        ta = torch.tensor([again], device=dev if backend != "gloo" else "cpu", dtype=torch.float64)
        if pipe.active:
            dist.all_reduce(ta, op=dist.ReduceOp.MAX)
        sustained["timed_region_again"] = {"steps": args.steps, "ms_per_step": 1e3 * float(ta.item()) / args.steps,
                                           "clips_per_s_per_gpu": B * args.steps / float(ta.item())}

    # the float64 log-mel kernel on the whole batch (what auto mode costs per clip it redoes), outside the timed region
    k1_f64_ms = None
    if rank == 0:
        ops.set_logmel_math("f64")
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for i in range(5):
            if i == 2: evs[0].record(stream)
            nat.check(nat.lib.ww_logmel_f32(p(pcm), B, 16000, 16000, 1, p(mel), st))
        evs[1].record(stream)
        torch.cuda.synchronize()
        k1_f64_ms = evs[0].elapsed_time(evs[1]) / 3
        ops.set_logmel_math(logmel_math)
        nat.check(nat.lib.ww_logmel_f32(p(pcm), B, 16000, 16000, 1, p(mel), st))     # leave the measured mode's mel behind
        torch.cuda.synchronize()

    # K1 on a NOISE-FREE batch: auto mode redoes clips with live bands on the float FFT's rounding floor in float64, which the
    # headline's sine + noise clips never trigger -- clean speech / synthetic tones do.  Tones, tone pairs, chirps and gated tones, 16-bit
    # quantised like a PCM-16 file (4096 distinct signals); the redone fraction = clips whose auto output differs from the f32 kernel's.
    k1_clean = None
    if rank == 0 and B >= 64:
        tt = np.arange(16000, dtype=np.float64) / 16000.0
        idx = np.arange(B)
        f0 = 110.0 * 2.0 ** ((idx % 61) / 12.0)
        sig = 0.5 * np.sin(2 * np.pi * f0[:, None] * tt[None, :])
        sig += np.where((idx % 3 == 1)[:, None], 0.25 * np.sin(2 * np.pi * (2.5 * f0)[:, None] * tt[None, :]), 0.0)
        sig *= np.where((idx % 5 == 2)[:, None], (tt[None, :] > 0.3), 1.0)
        sig = (np.round(np.clip(sig, -1, 1 - 2.0 ** -15) * 32768.0) / 32768.0).astype(np.float32)
        pcm_c = torch.from_numpy(sig).to(dev)
        mel_a, mel_f = torch.empty_like(mel), torch.empty_like(mel)
        times = {}
        for mode, dst in (("f32", mel_f), ("auto", mel_a)):
            ops.set_logmel_math(mode)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for i in range(7):
                if i == 2: evs[0].record(stream)
                nat.check(nat.lib.ww_logmel_f32(p(pcm_c), B, 16000, 16000, 1, p(dst), st))
            evs[1].record(stream)
            torch.cuda.synchronize()
            times[mode] = evs[0].elapsed_time(evs[1]) / 5
        ops.set_logmel_math(logmel_math)
        redone = float((mel_a != mel_f).flatten(1).any(dim=1).float().mean())
        k1_clean = {"workload": f"{B} noise-free 16-bit-quantised signals (tones 110 Hz .. 3.7 kHz, tone pairs, gated tones)",
                    "auto_ms": times["auto"], "f32_ms": times["f32"], "float64_redo_fraction": redone,
                    # what a digitally clean corpus costs end to end: K1 in auto mode on THIS batch + the timed region's K2 and K3
                    "e2e_noise_free_clips_per_s": B / ((times["auto"] + k2_ms + k3_ms) * 1e-3),
                    "note": "auto = the float32 kernel + the float64 kernel on the clips it marked; the headline's sine + noise clips mark none"}
        del pcm_c, mel_a, mel_f

    # the exact-f32 MFMA conv kernel, timed outside the timed region for the second roofline line
    k2_f32_ms = None
    if rank == 0 and conv_math != "f32":
        ops.set_conv_math("f32")
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for i in range(7):
            if i == 2: evs[0].record(stream)
            nat.check(nat.lib.ww_cnn_pool_f32(p(mel), B, 32, p(packed), n_conv, p(scratch) if scratch_bytes else None, p(pooled), st))
        evs[1].record(stream)
        torch.cuda.synchronize()
        k2_f32_ms = evs[0].elapsed_time(evs[1]) / 5
        ops.set_conv_math(conv_math)

    if rank == 0:
        clips_per_s = world * B * args.steps / elapsed
        k2_flops = K2_FLOPS_PER_CLIP[args.arch] * B
        k2_ach = k2_flops / (k2_ms * 1e-3)
        split = conv_math in ("f16x3", "f16x3d")
        wino = conv_math == "f16x3"            # conv2 (and conv3 of the 3-conv model) as 1-D Winograd F(2,3)
        # matrix-pipe flops actually issued per algorithmic flop: 3 (split) x 2/3 for conv2 as Winograd F(2,3) along rows
        issue_mult = (3.0 * (2.0 / 3.0 if wino else 1.0)) if split else 1.0
        k2_peak = MFMA_F16_PEAK if split else MFMA_F32_PEAK
        k2_kernels = (("cnn2w_kernel" if wino else "cnn2h16_kernel" if split else "cnn2_kernel") if args.arch == "simple" else
                      ("cnn2w_kernel+cnn3w_kernel" if wino else "cnn2h16_kernel+cnn3h_kernel" if split else "cnn2_kernel+cnn3_kernel"))
        k2_traffic, k2_traffic_src = pmc_traffic(k2_kernels, B, args.arch, with_source=True)
        k1_traffic, k1_traffic_src = pmc_traffic("logmel_kernel", B, args.arch, with_source=True)
        n_dev = torch.cuda.device_count()
        if backend == "gloo" and world > 1:
            exchange = f" -> gloo REHEARSAL of the logits all-gather, host-staged: {world} ranks on {min(world, n_dev)} GPU(s), not an RCCL measurement"
        elif pipe.active:
            exchange = f" -> RCCL all-gather of logits ({world}-rank group)"
        else:
            exchange = ""
        out = {
            "metric": METRIC, "value": clips_per_s, "unit": "clips/s", "n_gpus": (world if not (backend == "gloo" and world > 1) else min(world, n_dev)), "ranks": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 in/out/accumulate; conv and LSTM-gate GEMM products as f16x3 split (3 f16 MFMAs per fp32 product block, ~2^-21 rel.)" if split else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[2]: batch={B} 1 s/16 kHz clips per GPU, full log-mel + CNN + LSTM HIP forward "
                            f"(K1 -> K2 -> K3{exchange}), {B} distinct synthetic clips per GPU (seed = clip index), PCM and logits resident in HBM; "
                            + ("SimpleWakewordModel (train_wakeword.py:28-49)" if args.arch == "simple"
                               else "3-conv WakewordModel (wakeword_training_script.py:141-184)") + ", random-init weights seed 1234",
                "global_batch": world * B, "clips_per_gpu": B, "conv_math": conv_math, "logmel_math": logmel_math,
                "parallelism": (f"clips sharded over {world} GPU(s), replicated weights" if not (backend == "gloo" and world > 1) else
                                f"gloo rehearsal: {world} ranks sharing {min(world, n_dev)} GPU(s), replicated weights"),
                "backend": (backend if pipe.active else None),
            },
            "roofline": {
                "kernel": ("cnn2w_kernel (conv1 + conv2 + ReLU + avg-pool; conv2 as 1-D Winograd F(2,3) along rows, split-precision v_mfma_f32_16x16x32_f16 x3)" if wino else
                           "cnn2h16_kernel (conv1 + conv2 + ReLU + avg-pool; split-precision v_mfma_f32_16x16x32_f16 x3)" if split else
                           "cnn2_kernel<POOL> (conv1 + conv2 + ReLU + avg-pool, v_mfma_f32_32x32x2_f32)") if args.arch == "simple"
                          else ("cnn2w_kernel<false> + cnn3w_kernel (conv stack of the 3-conv model, both convs as 1-D Winograd, float32 intermediate in HBM)" if wino else
                                "cnn2h16_kernel<false> + cnn3h_kernel (direct split-precision convs, f16 hi/lo intermediate in HBM)" if split else
                                "cnn2_kernel<false> + cnn3_kernel (exact f32)"),
                "bound": "mfma", "achieved": k2_ach / 1e12, "peak": k2_peak / 1e12, "unit": "TFLOP/s",
                "frac": k2_ach / k2_peak, "traffic": k2_traffic, "traffic_source": k2_traffic_src,
                "flops_per_launch": k2_flops, "avg_launch_ms": k2_ms,
                "note": (f"achieved = ALGORITHMIC fp32 flops of the direct convolution; the kernel issues {issue_mult:.1f} f16 MFMA flops per "
                         "algorithmic flop (3 per product block for fp32-level accuracy"
                         + (", x 2/3 for conv2 as Winograd F(2,3))" if wino else ")")) if split else "exact fp32 products and accumulation",
                "mfma_issue_frac": issue_mult * k2_ach / k2_peak,
            },
            "stages": {
                "K1_logmel": {"avg_ms": k1_ms, "bound": "hbm", "achieved_GBps": K1_BYTES_PER_CLIP * B / (k1_ms * 1e-3) / 1e9,
                              "peak_GBps": HBM_PEAK / 1e9, "frac": K1_BYTES_PER_CLIP * B / (k1_ms * 1e-3) / HBM_PEAK,
                              "clips_per_s": B / (k1_ms * 1e-3), "traffic": k1_traffic, "traffic_source": k1_traffic_src,
                              "f32_vector_frac": K1_FLOPS_PER_CLIP * B / (k1_ms * 1e-3) / MFMA_F32_PEAK,
                              "math": logmel_math + (" (float32 FFT kernel + the launch that redoes marked clips in float64; none of the "
                                                     "benchmark's sine+noise clips is marked)" if logmel_math == "auto" else ""),
                              "f64_mode_ms_whole_batch": k1_f64_ms, "noise_free_batch": k1_clean},
                "K2_cnn": {"avg_ms": k2_ms, "clips_per_s": B / (k2_ms * 1e-3)},
                "K3_lstm_fc": dict({"avg_ms": k3_ms, "bound": "latency (one workgroup's layer 0 -> layer 1 -> fc chain; same time at 16 and 4096 clips)",
                                    "achieved_TFLOPs": K3_FLOPS_PER_CLIP[args.arch] * B / (k3_ms * 1e-3) / 1e12,
                                    "mfma": "v_mfma_f32_16x16x32_f16 x3 (split precision)" if split else "v_mfma_f32_16x16x4_f32",
                                    # the gate GEMMs against the peak of the instruction they run on (algorithmic flops; x3 issued when split)
                                    "mfma_frac_of_peak": K3_FLOPS_PER_CLIP[args.arch] * B / (k3_ms * 1e-3) / (MFMA_F16_PEAK if split else MFMA_F32_PEAK),
                                    "mfma_peak_TFLOPs": (MFMA_F16_PEAK if split else MFMA_F32_PEAK) / 1e12},
                                   **pmc_busy("lstm_fc", B, args.arch)),
                "kernel_ms_sum": k1_ms + k2_ms + k3_ms,
            },
            "device": nat.device_info(),
            "sync_timeouts": int(nat.lib.ww_sync_timeouts()),      # bounded in-kernel waits that expired: must be 0
        }
        if gather_us is not None:
            out["gather"] = {"collective": "all_gather_into_tensor of [B, 2] float32 logits per rank (RCCL)", "ranks": world,
                             "bytes_per_rank": B * 8, "blocking_latency_us": gather_us,
                             "note": "in the timed step the gather is asynchronous and overlaps the next step's kernels"}
        if sustained is not None:
            sustained["clips_per_s"] = sustained.pop("clips_per_s_per_gpu") * world
            sustained["timed_region_again"]["clips_per_s"] = sustained["timed_region_again"].pop("clips_per_s_per_gpu") * world
            out["sustained"] = sustained
        if k2_f32_ms is not None:
            out["roofline_f32_exact"] = {
                "kernel": "cnn2_kernel<POOL> (same stage with WW_CONV_MATH=f32: v_mfma_f32_32x32x2_f32, exact fp32)",
                "bound": "mfma", "achieved": k2_flops / (k2_f32_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": k2_flops / (k2_f32_ms * 1e-3) / MFMA_F32_PEAK, "avg_launch_ms": k2_f32_ms,
                "end_to_end_clips_per_s_with_it": B / ((k1_ms + k2_f32_ms + k3_ms) * 1e-3),
            }
        if world == 1 and not args.no_cpu_baseline:
            base, ref_mel, ref_logits = cpu_baseline(host, sd, budget_s=20.0)
            n = len(ref_mel)
            with torch.no_grad():
                got_mel = ops.logmel(pcm[:n], True).cpu().numpy()
                got_logits = model.forward_pcm(pcm[:n]).cpu().numpy()
            out["cpu_baseline"] = base
            out["parity"] = {"clips": n, "logmel_max_abs_err_dB": float(np.abs(got_mel - ref_mel).max()),
                             "logits_max_abs_err": float(np.abs(got_logits - ref_logits).max()),
                             "against": "oracle/ (librosa is not installed: mel parity vs librosa itself is unpinned)"}
            out["gpu_over_cpu"] = clips_per_s / base["value"]
        if world == 1 and not args.no_streaming and args.arch == "simple":
            # BASELINE configs[4]: 256 microphones, 10 ms hop, one hipGraph replay per hop (after the timed region)
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import bench_streaming
            out["streaming"] = bench_streaming.measure(mics=256, hop=160, hops=1000, device=dev.index)
            # AudioProcessor.augment_audio on the GPU (training-side stage that feeds K1)
            import bench_augment
            out["augmentation"] = bench_augment.measure(batch=B, steps=5, check=4, device=dev.index)
            # one training step of the same model (forward in train mode + backward + Adam)
            import bench_train
            out["training"] = bench_train.measure(batch=B, steps=20, device=dev.index)
            out["training"]["exact_fp32_ms_per_step"] = bench_train.measure(batch=B, steps=3, device=dev.index, math="f32", cpu_sample=8)["ms_per_step"]
            bench_train.ops_reset_train_math()
            out["training_pipeline"] = bench_train.measure_pipeline(batch=B, steps=3, device=dev.index)
            out["training_3conv"] = bench_train.measure(batch=2048, steps=10, device=dev.index, arch="full", cpu_sample=16)
            out["training_3conv"]["exact_fp32_ms_per_step"] = bench_train.measure(batch=2048, steps=2, device=dev.index, arch="full", math="f32", cpu_sample=4)["ms_per_step"]
            bench_train.ops_reset_train_math()
            # the notebook's model in inference (driver-run figure for SURVEY A11 / 8(f).3's forward half)
            import bench_forward
            out["forward_3conv"] = bench_forward.measure(batch=B, steps=10, device=dev.index, pcm=pcm)
            # K0 alone, and WAV files -> logits
            import bench_files
            out["decode"] = bench_files.measure_decode(batch=B, steps=10, device=dev.index)
            out["file_pipeline"] = bench_files.measure_file_pipeline(n_files=4096, batch=1024, passes=256, device=dev.index)   # one stream of 1 M files (~1 s: ten quota periods)
        print(json.dumps(out))
    if pipe.active:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
