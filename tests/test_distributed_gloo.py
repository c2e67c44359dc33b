"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): sharding + the one collective (logits all-gather).

The model itself has no CPU implementation, so each rank fabricates its shard's logits with the oracle; what is
under test is the N > 1 plumbing bench.py and sharded_forward_pcm use: init_from_env, shard_bounds,
all_gather_logits (equal and ragged shards), rank-order of the result.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    # CPU ranks: they must not open a GPU that happens to be there (a GPU box allows six processes on its card; this file runs eight)
    os.environ.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from wakeword_jupyterlab_amd import distributed as wd

    r, w, _ = wd.init_from_env("gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    lo, hi = wd.shard_bounds(n_total, rank, world)
    # "logits" of clip i are (i, -i): any misplacement shows up in the gathered tensor
    local = torch.stack([torch.arange(lo, hi, dtype=torch.float32), -torch.arange(lo, hi, dtype=torch.float32)], 1)
    full = wd.all_gather_logits(local, n_total)
    full_nohint = wd.all_gather_logits(local)              # without n_total: padded to the largest shard
    # the bench step's exchange: double-buffered gathers, five steps, step k's logits = (rank, k)
    pipe = wd.LogitsGatherPipeline(3, "cpu")
    assert pipe.active and pipe.world == world and pipe.backend == "gloo"
    seen = []
    for k in range(5):
        buf = pipe.acquire()
        buf[:, 0], buf[:, 1] = float(rank), float(k)
        seen.append(pipe.submit())
    pipe.drain()
    for k in (3, 4):                                       # the two live buffer pairs hold the last two steps
        assert torch.equal(seen[k][:, 1], torch.full((world * 3,), float(k)))
        assert torch.equal(seen[k][:, 0], torch.arange(world, dtype=torch.float32).repeat_interleave(3))
    q.put((rank, lo, hi, full.numpy(), full_nohint.shape[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 1])
def test_two_rank_all_gather_of_logits(n_total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([np.arange(n_total, dtype=np.float32), -np.arange(n_total, dtype=np.float32)], 1)
    covered = []
    for rank, lo, hi, full, n_nohint in results:
        assert np.array_equal(full, want)                  # every rank holds every clip's logits, in clip order
        assert n_nohint == world * ((n_total + world - 1) // world)
        covered += list(range(lo, hi))
    assert sorted(covered) == list(range(n_total))         # shards partition the batch


@pytest.mark.parametrize("n_total", [32768 // 64, 509])
def test_eight_rank_all_gather_and_pipeline(n_total):
    """BASELINE configs[3]'s rank count (8 ranks, one per GPU) on CPU: equal shards (512 = 8 x 64) and ragged ones (509: the last rank is
    short), the padded no-hint form, and the bench step's double-buffered LogitsGatherPipeline at world size 8.  No 8-GPU node was ever
    available to this build; this is what can be checked without one."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = np.stack([np.arange(n_total, dtype=np.float32), -np.arange(n_total, dtype=np.float32)], 1)
    covered = []
    for rank, lo, hi, full, n_nohint in results:
        assert np.array_equal(full, want)
        assert n_nohint == world * ((n_total + world - 1) // world)
        assert hi - lo == (64 if n_total == 512 else (64 if rank < 7 else 509 - 7 * 64))
        covered += list(range(lo, hi))
    assert sorted(covered) == list(range(n_total))


def test_reader_threads_follow_the_cpu_share_of_a_rank(monkeypatch):
    """A node's ranks share its CPUs: the file reader's thread count is derived from the affinity mask, the cgroup quota and
    LOCAL_WORLD_SIZE, not a constant."""
    from wakeword_jupyterlab_amd import files
    monkeypatch.delenv("WW_READER_THREADS", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    one = files.host_cpu_share()
    assert 1 <= one <= len(os.sched_getaffinity(0))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert files.host_cpu_share() == max(1, one // 8)
    share = max(1, one // 8)
    assert files.default_threads() == max(1, min(32, share - 1 if share >= 8 else share))
    monkeypatch.setenv("WW_READER_THREADS", "5")
    assert files.default_threads() == 5


def test_shard_bounds_partition_any_batch():
    from wakeword_jupyterlab_amd.distributed import shard_bounds
    for n in (0, 1, 5, 4096, 32768, 32769):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) == (n + world - 1) // world
    assert [shard_bounds(32768, r, 8) for r in (0, 7)] == [(0, 4096), (28672, 32768)]   # BASELINE configs[3]


def test_single_process_is_a_no_op():
    from wakeword_jupyterlab_amd.distributed import all_gather_logits, init_from_env
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert init_from_env("gloo") == (0, 1, 0)
    x = torch.randn(5, 2)
    assert all_gather_logits(x) is x


def _forced_single(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from wakeword_jupyterlab_amd import distributed as wd
    assert wd.init_from_env("gloo", force=True) == (0, 1, 0) and dist.is_initialized()
    x = torch.randn(5, 2)
    y = wd.all_gather_logits(x, 5)                         # a 1-rank group still runs the collective
    pipe = wd.LogitsGatherPipeline(4, "cpu")
    pipe.acquire().fill_(7.0)
    g = pipe.submit()
    pipe.drain()
    q.put((y is not x and torch.equal(x, y), bool((g == 7.0).all()) and g.shape == (4, 2) and pipe.active))
    dist.destroy_process_group()


def test_forced_one_rank_group_runs_the_collective():
    """What tests/_rccl_child.py does with RCCL on the GPU box, rehearsed here on gloo: WORLD_SIZE=1 with a real group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_single, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120) == (True, True)
    p.join(timeout=60)
    assert p.exitcode == 0
