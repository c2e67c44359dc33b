"""Why is K2 slower inside the step (behind K1) than looped alone?  Times K2 in several contexts with HIP events."""
import ctypes as C, numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops, _native as nat
dev = torch.device("cuda", 0)
B = 4096
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev)
packed = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict("simple"))).to(dev)
mel = torch.empty((B, 1, 80, 32), device=dev); pooled = torch.empty((B, 64), device=dev); logits = torch.empty((B, 2), device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
k1 = lambda: nat.check(nat.lib.ww_logmel_f32(p(pcm), B, 16000, 16000, 1, p(mel), st))
k2 = lambda: nat.check(nat.lib.ww_cnn_pool_f32(p(mel), B, 32, p(packed), 2, None, p(pooled), st))
k3 = lambda: nat.check(nat.lib.ww_lstm_fc_f32(p(pooled), B, p(packed), 2, p(logits), None, st))
def timed(seq, reps=30):
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)] for _ in range(reps)]
    for _ in range(5):
        for f in seq: f()
    torch.cuda.synchronize()
    for r in range(reps):
        evs[r][0].record()
        for i, f in enumerate(seq):
            f(); evs[r][i + 1].record()
    torch.cuda.synchronize()
    return [float(np.median([evs[r][i].elapsed_time(evs[r][i + 1]) for r in range(reps)])) for i in range(len(seq))]
print("K2 alone            ", ["%.4f" % t for t in timed([k2])])
print("K1 K2 K3            ", ["%.4f" % t for t in timed([k1, k2, k3])])
print("K1 K2 K2 K3         ", ["%.4f" % t for t in timed([k1, k2, k2, k3])])
print("K3 K2               ", ["%.4f" % t for t in timed([k3, k2])])
print("K1 K1 K2            ", ["%.4f" % t for t in timed([k1, k1, k2])])
