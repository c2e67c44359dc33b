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


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*; returns (rank, world, local_rank)."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", rank))
    if world > 1 and not dist.is_initialized():
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
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_logits
    world = dist.get_world_size(group)
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
