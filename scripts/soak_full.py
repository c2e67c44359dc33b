"""Which kernel of the 3-conv stack is non-deterministic?  Fixed scratch, compare scratch (conv2 output) and pooled bitwise."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat, ops
dev = torch.device("cuda", 0)
B = 777
packed = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict("full", seed=3))).to(dev)
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev)
mel = ops.logmel(pcm, True).clone()
scratch = torch.zeros(nat.lib.ww_cnn_scratch_bytes(B, 3), dtype=torch.uint8, device=dev)
pooled = torch.empty(B, 128, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
def run(fill=None):
    if fill is not None: scratch.fill_(fill)
    nat.check(nat.lib.ww_cnn_pool_f32(p(mel), B, 32, p(packed), 3, p(scratch), p(pooled), None))
    torch.cuda.synchronize()
for mode in ("f16x3", "f32"):
    ops.set_conv_math(mode)
    run(0); ref_s, ref_p = scratch.clone(), pooled.clone()
    bad_s = bad_p = 0; n = 0; t0 = time.time(); fills = [0, 255, 0x7b]
    while time.time() - t0 < 30:
        run(fills[n % 3]); n += 1
        if not torch.equal(scratch, ref_s):
            bad_s += 1
            if bad_s <= 3:
                d = (scratch != ref_s).nonzero().flatten()
                print(mode, "scratch differs at", d[:6].tolist(), "count", d.numel(), "fill", fills[(n - 1) % 3])
        if not torch.equal(pooled, ref_p): bad_p += 1
    print(mode, {"runs": n, "scratch_mismatch": bad_s, "pooled_mismatch": bad_p})
# same without touching the scratch between runs
ops.set_conv_math("f16x3")
run(0); ref_p = pooled.clone(); bad = 0
for i in range(3000):
    run()
    if not torch.equal(pooled, ref_p): bad += 1
print("f16x3, scratch left alone:", {"runs": 3000, "pooled_mismatch": bad})
