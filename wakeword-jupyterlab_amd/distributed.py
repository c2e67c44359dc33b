"""Batch-sharded inference across the GPUs of a node: one process per GPU, RCCL over xGMI.

The reference has no multi-GPU code (a README-only `nn.DataParallel` snippet, README.md:247-253).  Clips
are independent (per-clip normalisation, per-clip dB max, eval-mode model), so the batch shards with no
data-path exchange; the one collective is an all-gather of the per-clip logits ([B/R, 2] f32 per rank,
32 KiB at 4096 clips) so that every rank -- in particular the host loop that computes the metrics -- sees
the whole batch.  `backend='nccl'` IS RCCL on ROCm; the same code runs on `gloo` with CPU tensors for the
host-logic tests (the gather only; the model itself has no CPU path).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None, force: bool = False) -> tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*; returns (rank, world, local_rank).
    A single rank does not need a group; `force=True` creates one anyway (a 1-rank RCCL communicator: the collective
    path of this module can then be executed on a one-GPU box)."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", rank))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Rank r of R owns clips [r*ceil(n/R), min(n, (r+1)*ceil(n/R))): contiguous, sizes differ by at most the tail."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def all_gather_logits(local_logits: torch.Tensor, n_total: int | None = None, group=None) -> torch.Tensor:
    """[B_rank, 2] on every rank -> [sum B_rank, 2] on every rank, in rank order.

    Equal shards use one `all_gather_into_tensor` (a single ncclAllGather); ragged shards (last rank short)
    are padded to the largest shard first and trimmed after, so the collective stays a single call."""
    if not dist.is_initialized():
        return local_logits
    world = dist.get_world_size(group)          # a 1-rank group still goes through the collective (a copy): same code at every N
    n_local = local_logits.shape[0]
    if n_total is not None:
        per = (n_total + world - 1) // world
    else:
        gloo = dist.get_backend(group) == "gloo"
        t = torch.tensor([n_local], device="cpu" if gloo else local_logits.device, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        per = int(t.item())
    send = local_logits.contiguous()
    if n_local != per:
        send = torch.zeros((per,) + tuple(local_logits.shape[1:]), device=local_logits.device, dtype=local_logits.dtype)
        send[:n_local] = local_logits
    out = torch.empty((world * per,) + tuple(local_logits.shape[1:]), device=local_logits.device, dtype=local_logits.dtype)
    if send.device.type == "cuda" and dist.get_backend(group) == "gloo":
        # rehearsal on a box with fewer GPUs than ranks (ranks share a GPU, gloo has no device path): stage through the host
        parts = [torch.empty(send.shape, dtype=send.dtype) for _ in range(world)]
        dist.all_gather(parts, send.cpu(), group=group)
        out.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(out, send, group=group)
    if n_total is not None and n_total != world * per:
        out = out[:n_total]            # only the last ranks can be short with contiguous ceil-sharding
    return out


def sharded_forward_pcm(model, pcm_local: torch.Tensor, n_total: int | None = None, normalize: bool = True, group=None):
    """Each rank runs PCM -> logits on its own shard, then all ranks receive the full [N, 2] logits."""
    with torch.no_grad():
        local = model.forward_pcm(pcm_local, normalize)
    return all_gather_logits(local, n_total, group)


class LogitsGatherPipeline:
    """The per-step exchange of the batched inference loop: a double-buffered, asynchronous all-gather of [B, 2] logits.

    Step k writes its logits into `acquire()`'s buffer and calls `submit()`; the gather then runs on the communicator's own
    stream (RCCL) while step k+1 computes into the other buffer pair.  `acquire()` first waits for the gather that used this
    pair two steps ago.  Without a process group the "gathered" tensor is the local one and nothing is launched.
    bench.py's timed step and the GPU test of the RCCL path (tests/_rccl_child.py) both drive exactly this class."""

    def __init__(self, batch: int, device, group=None, depth: int = 2):
        self.group = group
        self.active = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.backend = dist.get_backend(group) if self.active else None
        self.batch = batch
        self.logits = [torch.empty((batch, 2), device=device, dtype=torch.float32) for _ in range(depth)]
        self.gathered = ([torch.empty((self.world * batch, 2), device=device, dtype=torch.float32) for _ in range(depth)]
                         if self.active else self.logits)
        self.pending = [None] * depth
        self.count = 0
        self.slot = 0

    def acquire(self) -> torch.Tensor:
        """The logits buffer of the next step (its previous gather, if any, has completed)."""
        self.slot = self.count % len(self.logits)
        self.count += 1
        self._wait(self.slot)
        return self.logits[self.slot]

    def submit(self) -> torch.Tensor:
        """Issue the gather of the buffer `acquire()` returned; returns the tensor that will hold all ranks' logits."""
        b = self.slot
        if self.active:
            if self.backend == "gloo":      # rehearsal: ranks may share a GPU and gloo has no device path -> through the host
                parts = [torch.empty((self.batch, 2)) for _ in range(self.world)]
                dist.all_gather(parts, self.logits[b].cpu(), group=self.group)
                self.gathered[b].copy_(torch.cat(parts))
            else:
                self.pending[b] = dist.all_gather_into_tensor(self.gathered[b], self.logits[b], group=self.group, async_op=True)
        return self.gathered[b]

    def _wait(self, b: int) -> None:
        if self.pending[b] is not None:
            self.pending[b].wait()          # makes the current stream wait for the collective; does not block the host
            self.pending[b] = None

    def drain(self) -> None:
        """Wait for every outstanding gather (then barrier + synchronise outside, as the bench contract asks)."""
        for b in range(len(self.pending)):
            self._wait(b)

    def latest(self) -> torch.Tensor:
        return self.gathered[self.slot]
