"""Reader threads vs file-fed rate on the GPU box: what the host really grants (affinity mask, cgroup cpu.max), then
`bench_files.measure_file_pipeline` for several thread counts, interleaved twice; each run is one stream of SWEEP_PASSES (128) x 4096 files -- many quota periods -- in a
temporary directory under SWEEP_TMP_ROOT (default: the system's).  usage: PYTHONPATH=. python scripts/sweep_reader_threads.py [t1,t2,...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_files

counts = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "8,16,24,32,48,64".split(","))]
info = {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        info[f] = open(f).read().strip()
    except OSError:
        pass
res = {"host": info, "runs": []}
print(json.dumps(info), flush=True)
for rnd in range(int(os.environ.get("SWEEP_ROUNDS", "2"))):
    for t in counts:
        r = bench_files.measure_file_pipeline(passes=int(os.environ.get("SWEEP_PASSES", "128")), threads=t, tmp_root=os.environ.get("SWEEP_TMP_ROOT"))
        row = {"round": rnd, "threads": t, "seconds": round(r["seconds"], 2), "host_cpus_used": r["host_cpus_used"], "quota_periods_throttled": r["quota_periods_throttled"], "clips_per_s": r["clips_per_s"], "host_read_only_clips_per_s": r["host_read_only_clips_per_s"],
               "wait_ms": r["main_thread_ms_waiting_for_reader_per_batch"], "enqueue_ms": r["main_thread_ms_enqueue_per_batch"]}
        res["runs"].append(row)
        print(json.dumps(row), flush=True)
print("RESULT " + json.dumps(res))
