"""The RCCL leg of tests/test_gpu_callers.py::test_rccl_gather_path_one_rank (not a test module itself).

A fresh process creates a ONE-rank `nccl` (= RCCL) process group on GPU 0 and drives exactly what bench.py's timed step and
`sharded_forward_pcm` drive at N > 1: `LogitsGatherPipeline` (async `all_gather_into_tensor` into double-buffered outputs,
`wait` two steps later) around K1 -> K2 -> K3, and `all_gather_logits`.  A 1-rank all-gather is a device copy issued by
RCCL on its own stream: the communicator, the stream ordering (`Work.wait`) and the buffer rotation are the real ones.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import wakeword_jupyterlab_amd as pkg  # noqa: E402
from wakeword_jupyterlab_amd import distributed as wdist  # noqa: E402


def main():
    rank, world, local = wdist.init_from_env("nccl", force=True)
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", local)
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    model = pkg.SimpleWakewordModel()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    B, steps = 96, 7
    clips = torch.from_numpy(pkg.synth.make_clips(0, B * 2)).to(dev)
    with torch.no_grad():
        want = [model.forward_pcm(clips[:B]), model.forward_pcm(clips[B:])]        # two different batches, alternating
        torch.cuda.synchronize()
        pipe = wdist.LogitsGatherPipeline(B, dev)
        assert pipe.active and pipe.backend == "nccl" and pipe.world == 1
        outs = []
        for k in range(steps):
            buf = pipe.acquire()
            buf.copy_(model.forward_pcm(clips[(k & 1) * B:(k & 1) * B + B]))
            outs.append(pipe.submit())
            assert pipe.pending[pipe.slot] is not None                              # the gather really is asynchronous work
        pipe.drain()
        dist.barrier()
        torch.cuda.synchronize()
        for k in (steps - 2, steps - 1):                                            # the two live buffer pairs
            assert outs[k].data_ptr() != pipe.logits[k % 2].data_ptr()
            assert torch.equal(outs[k], want[k & 1]), float((outs[k] - want[k & 1]).abs().max())
        # sharded_forward_pcm over RCCL: the 1-rank gather of a ragged request (n_total < padded size cannot occur at N = 1)
        g = wdist.sharded_forward_pcm(model, clips[:B], n_total=B)
        assert g.data_ptr() != want[0].data_ptr() and torch.equal(g, want[0])
        g2 = wdist.all_gather_logits(want[1])                                       # size agreed by an all-reduce on the device
        assert torch.equal(g2, want[1])
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_OK")


if __name__ == "__main__":
    main()
