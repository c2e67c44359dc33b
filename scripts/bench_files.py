"""File-fed legs of the bench line (SURVEY 8(f).1 -- the reference's own published figure, 453 clips/s, wakeword_training.ipynb:742,
is a file-fed rate):
  decode         K0 alone on sample bytes already in HBM (16 kHz s16 mono, and 48 kHz s16 mono through the polyphase resampler)
  file_pipeline  WAV files on disk -> logits: written to a temporary directory by create_sample_data's recipe
                 (wakeword_training_script.py:350-393: 1 s, 16 kHz, PCM-16 like soundfile's default), read by the library's host
                 threads into pinned staging, H2D on the copy stream, K0 -> K1 -> K2 -> K3, double-buffered
Alone: PYTHONPATH=. python scripts/bench_files.py"""
import ctypes as C
import json
import os
import shutil
import struct
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _wav_bytes(x, sr=16000):
    raw = np.clip(np.round(np.asarray(x, np.float64) * 32767), -32768, 32767).astype("<i2").tobytes()
    return b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16) \
        + b"data" + struct.pack("<I", len(raw)) + raw


def measure_decode(batch=4096, steps=10, device=0):
    """K0 alone, bytes resident in HBM."""
    import wakeword_jupyterlab_amd as pkg
    from wakeword_jupyterlab_amd import _native as nat
    dev = torch.device("cuda", device)
    out = {}
    for sr in (16000, 44100, 48000):
        frames = sr
        base = (pkg.synth.make_clip(3)[: min(16000, frames)] * 20000).astype("<i2")
        one = np.resize(base, frames).astype("<i2").tobytes()
        one += b"\0" * (-len(one) % 16)
        raw = torch.frombuffer(bytearray(one * batch), dtype=torch.uint8).to(dev)
        proto = nat.ClipDesc()
        with torch.cuda.device(dev):
            nat.check(nat.lib.ww_resampler_prepare(sr, C.byref(proto)))
        descs = (nat.ClipDesc * batch)()
        for i, d in enumerate(descs):
            d.byte_offset, d.n_frames, d.channels, d.sample_rate, d.format = i * len(one), frames, 1, sr, nat.FMT_S16
            d.up, d.down, d.half_len, d.taps_dev, d.crop_start = proto.up, proto.down, proto.half_len, proto.taps_dev, 0
        desc_t = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        pcm = torch.empty((batch, 16000), device=dev)
        stream = torch.cuda.current_stream()
        st = C.c_void_p(stream.cuda_stream)
        call = lambda: nat.check(nat.lib.ww_decode_resample(C.c_void_p(raw.data_ptr()), C.c_void_p(desc_t.data_ptr()), batch, 1,  # noqa: E731
                                                            C.c_void_p(pcm.data_ptr()), st))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(steps):
            call()
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / steps
        algo = (frames * 2 + 16000 * 4) * batch                 # sample bytes in + float32 clip out
        out[f"{sr // 1000}k_s16_mono"] = {"ms": ms, "clips_per_s": batch / (ms * 1e-3), "algorithmic_GBps": algo / (ms * 1e-3) / 1e9,
                                          "hbm_frac_of_8TBps": algo / (ms * 1e-3) / 8e12, "finite": bool(torch.isfinite(pcm).all())}
    out["workload"] = f"decode_resample_kernel (K0) alone, {batch} files of 1 s, sample bytes resident in HBM -> normalised float32 clips"
    return out


def _cgroup_cpu_stat():
    """usage_usec / nr_periods / nr_throttled / throttled_usec of this process's cgroup (v2), {} where there is none."""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            return {k: int(v) for k, v in (line.split() for line in f)}
    except (OSError, ValueError):
        return {}


def measure_file_pipeline(n_files=4096, batch=1024, passes=256, device=0, threads=None, tmp_root=None):
    """WAV files -> logits, end to end, page cache warm (the files were just written)."""
    import wakeword_jupyterlab_amd as pkg
    from wakeword_jupyterlab_amd import files as files_mod
    from wakeword_jupyterlab_amd.files import EncodedPaths, WavBatchReader, default_threads
    dev = torch.device("cuda", device)
    threads = threads or default_threads()
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = pkg.SimpleWakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    tmp = tempfile.mkdtemp(prefix="ww_bench_wavs_", dir=tmp_root)
    try:
        base = pkg.synth.make_clips(0, 64)
        base = base / np.abs(base).max(axis=1, keepdims=True) * 0.9
        paths = []
        for i in range(n_files):
            p = os.path.join(tmp, f"clip_{i:05d}.wav")
            with open(p, "wb") as f:
                f.write(_wav_bytes(base[i % 64] * (1.0 - 0.004 * (i // 64))))
            paths.append(p)
        file_bytes = os.path.getsize(paths[0])
        enc = EncodedPaths(paths)                    # the dataset's file list as the library takes it, converted once
        rd = WavBatchReader(max_clips=batch, max_raw_bytes=batch * (32000 + 64), threads=threads, slots=3, device=dev)
        n_b = n_files // batch
        logits = [torch.empty((batch, 2), device=dev) for _ in range(n_b)]

        def stream(enc_, timing=None):
            t_prev = time.perf_counter()
            for b, (buf, ok) in enumerate(rd.stream(enc_, batch, normalize=True, verbose=False)):     # reader thread one batch ahead
                t1 = time.perf_counter()
                with torch.no_grad():
                    logits[b % n_b].copy_(m.forward_pcm(buf, normalize=False))   # K0 already normalised over the whole file
                if timing is not None:
                    timing[0] += t1 - t_prev                                      # waiting for the reader + upload / K0 enqueue
                    timing[1] += time.perf_counter() - t1
                t_prev = time.perf_counter()
                assert ok.all()
        stream(enc)
        torch.cuda.synchronize(dev)
        first = torch.cat(logits).clone()
        # ONE stream over `passes` repetitions of the file list: an epoch is one long stream, and the host's CPU quota (cgroup cpu.max)
        # is enforced per 100 ms period -- a run shorter than a few periods measures a burst the quota has not caught up with yet
        long_enc = EncodedPaths(paths * passes)
        timing = [0.0, 0.0]
        cpu0 = _cgroup_cpu_stat()
        t0 = time.perf_counter()
        stream(long_enc, timing)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        cpu1 = _cgroup_cpu_stat()
        same = bool(torch.equal(first, torch.cat(logits)))
        # the host side alone (read into staging, no GPU work queued behind it), over the same number of files
        t1 = time.perf_counter()
        for b in range(n_b * passes):
            rd.read(long_enc, b % 3, b * batch, (b + 1) * batch)
        host_dt = (time.perf_counter() - t1) / passes
        rd.close()
        n = n_files * passes
        return {"workload": f"{n_files} WAV files (1 s, 16 kHz, PCM-16, {file_bytes} B each; create_sample_data's format) in a temporary directory, page cache warm "
                            f"-> logits: library reader threads (driven one batch ahead by a helper thread: WavBatchReader.stream) -> pinned staging -> H2D (copy stream) -> K0 -> K1 -> K2 -> K3, batches of {batch}, 3 staging slots, "
                            f"SimpleWakewordModel; one stream over {passes} repetitions of the file list ({n_files * passes} files, {dt:.2f} s: many periods of the host's CPU quota)",
                "seconds": dt, "host_cpus_used": (cpu1["usage_usec"] - cpu0["usage_usec"]) * 1e-6 / dt if cpu0 else None,
                "quota_periods_throttled": [cpu1["nr_throttled"] - cpu0["nr_throttled"], cpu1["nr_periods"] - cpu0["nr_periods"]] if cpu0 else None,
                "clips_per_s": n / dt, "ms_per_batch": 1e3 * dt / (n_b * passes), "host_threads": threads, "host_cpus": os.cpu_count(), "host_cpu_share_of_this_rank": files_mod.host_cpu_share(),
                "local_world_size": int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1),
                "host_read_only_clips_per_s": n_files / host_dt, "main_thread_ms_waiting_for_reader_per_batch": 1e3 * timing[0] / (n_b * passes),
                "main_thread_ms_enqueue_per_batch": 1e3 * timing[1] / (n_b * passes),
                "file_MBps": n * file_bytes / dt / 1e6, "bitwise_repeatable_across_passes": same,
                "reference_published_clips_per_s": 453, "reference_source": "wakeword_training.ipynb:742 (RTX 3060 Ti, DataLoader num_workers=2)"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    print(json.dumps({"decode": measure_decode(), "file_pipeline": measure_file_pipeline()}))
