"""Where a file-fed batch's time goes: the producer's read() calls and the consumer's queue wait / crop draw / decode() / forward enqueue,
stamped with perf_counter over one long stream (the 4096 files repeated).  usage: PYTHONPATH=. python scripts/proto/pipeline_trace.py [threads] [batch]"""
import json, os, shutil, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_files
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd.files import EncodedPaths, WavBatchReader, default_threads

threads = int(sys.argv[1]) if len(sys.argv) > 1 else default_threads()
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = 8
dev = torch.device("cuda", 0)
m = pkg.SimpleWakewordModel()
m.load_state_dict({k: torch.from_numpy(v) for k, v in pkg.synth.make_state_dict("simple", seed=1234).items()})
m = m.to(dev).eval()
tmp = tempfile.mkdtemp(prefix="ww_trace_")
try:
    base = pkg.synth.make_clips(0, 64)
    base = base / np.abs(base).max(axis=1, keepdims=True) * 0.9
    paths = []
    for i in range(4096):
        p = os.path.join(tmp, f"c{i:05d}.wav")
        with open(p, "wb") as f:
            f.write(bench_files._wav_bytes(base[i % 64]))
        paths.append(p)
    enc = EncodedPaths(paths * reps)
    rd = WavBatchReader(max_clips=batch, max_raw_bytes=batch * (32000 + 64), threads=threads, slots=3, device=dev)
    reads = []
    inner = rd.read
    def read(*a, **k):
        t = time.perf_counter(); r = inner(*a, **k); reads.append(time.perf_counter() - t); return r
    rd.read = read
    decs = []
    inner_d = rd.decode
    def decode(*a, **k):
        t = time.perf_counter(); r = inner_d(*a, **k); decs.append(time.perf_counter() - t); return r
    rd.decode = decode
    out = torch.empty((batch, 2), device=dev)
    for warm in (True, False):
        reads.clear(); decs.clear()
        fwd, gaps, stamps = [], [], []
        t0 = t_prev = time.perf_counter()
        for buf, ok in rd.stream(enc, batch, verbose=False):
            t1 = time.perf_counter()
            with torch.no_grad():
                out.copy_(m.forward_pcm(buf, normalize=False))
            t2 = time.perf_counter()
            gaps.append(t1 - t_prev); fwd.append(t2 - t1); t_prev = t2; stamps.append(t1 - t0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ms = lambda v: round(1e3 * float(np.median(v)), 3)
    alone = []
    for b in range(16):
        t = time.perf_counter(); inner(enc, b % 3, b * batch, (b + 1) * batch); alone.append(time.perf_counter() - t)
    print(json.dumps({"threads": threads, "batch": batch, "clips_per_s": round(len(enc) / dt), "ms_per_batch": round(1e3 * dt / len(gaps), 3),
                      "read_ms_in_pipeline_p50": ms(reads), "read_ms_in_pipeline_p90": round(1e3 * float(np.quantile(reads, 0.9)), 3),
                      "read_ms_alone_p50": ms(alone), "decode_call_ms_p50": ms(decs), "forward_enqueue_ms_p50": ms(fwd),
                      "consumer_gap_ms_p50": ms(gaps)}))
    g = np.asarray(gaps); big = np.nonzero(g > 3 * np.median(g))[0]
    print("stalls (batch, at ms, gap ms):", [(int(i), round(1e3 * stamps[i], 1), round(1e3 * g[i], 2)) for i in big][:40])
    r = np.asarray(reads); bigr = np.nonzero(r > 2 * np.median(r))[0]
    print("slow reads (call, ms):", [(int(i), round(1e3 * r[i], 2)) for i in bigr][:40])
    try:
        print("cpu.stat:", open("/sys/fs/cgroup/cpu.stat").read().replace("\n", " | "))
    except OSError as e:
        print("cpu.stat:", e)
    rd.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
