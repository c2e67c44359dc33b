"""One rank of tests/test_gpu_callers.py::test_sharded_forward_pcm_two_ranks_on_one_gpu (not a test module itself).

RANK / WORLD_SIZE / MASTER_* come from the environment; both ranks use GPU 0 of the box and rendezvous over gloo.
Every rank checks the gathered logits against a single-process forward_pcm of the WHOLE batch, clip for clip (bitwise:
the kernels are batch-position independent), and rank 0 also against the CPU oracle on a sample.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import wakeword_jupyterlab_amd as pkg  # noqa: E402
from wakeword_jupyterlab_amd import distributed as wdist  # noqa: E402


def main():
    n_total = int(sys.argv[1])
    rank, world, _ = wdist.init_from_env("gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    model = pkg.SimpleWakewordModel()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    clips = pkg.synth.make_clips_tiled(0, n_total, unique=48)
    lo, hi = wdist.shard_bounds(n_total, rank, world)
    assert (lo, hi) == ((0, 151) if rank == 0 else (151, 301))
    local = torch.from_numpy(clips[lo:hi]).to(dev)
    gathered = wdist.sharded_forward_pcm(model, local, n_total=n_total)
    assert gathered.shape == (n_total, 2) and gathered.device.type == "cuda"
    with torch.no_grad():
        whole = model.forward_pcm(torch.from_numpy(clips).to(dev))
    assert torch.equal(gathered, whole), float((gathered - whole).abs().max())
    # without n_total the shard sizes are agreed on by an all-reduce; the padded tail rows are then kept (documented)
    g2 = wdist.all_gather_logits(whole[lo:hi])
    assert g2.shape[0] == 2 * 151 and torch.equal(g2[:151], whole[:151]) and torch.equal(g2[151:301], whole[151:])
    if rank == 0:
        from oracle import mel_oracle, model_oracle
        ref = model_oracle.forward_np(mel_oracle.logmel_batch(clips[145:157], normalize=True), sd)   # straddles the shard boundary
        assert np.abs(gathered[145:157].cpu().numpy() - ref).max() <= 1e-3
    dist.barrier()
    dist.destroy_process_group()
    print("SHARDED_OK rank", rank)


if __name__ == "__main__":
    main()
