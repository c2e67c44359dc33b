"""K1 under its three arithmetics (ww_set_logmel_math): ms per 4096 clips on the headline's sine + noise clips and on a batch of noise-free
quantised signals (what auto mode redoes in float64), plus the error against the oracle on 48 noise-free clips.
usage: PYTHONPATH=. python scripts/k1_modes.py [lib.so ...]   (WW_LIB_OVERRIDE-style A/B: each library is loaded into its own subprocess)"""
import json, os, subprocess, sys
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ, WW_LIB_OVERRIDE=os.path.abspath(lib))
        p = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(os.path.basename(lib), p.stdout.strip().splitlines()[-1] if p.returncode == 0 else p.stderr[-800:])
    sys.exit(0)

import torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops
from oracle import mel_oracle

dev = torch.device("cuda", 0)
B = 4096
noisy = torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev)
tt = np.arange(16000, dtype=np.float64) / 16000.0
idx = np.arange(B)
f0 = 110.0 * 2.0 ** ((idx % 61) / 12.0)
sig = 0.5 * np.sin(2 * np.pi * f0[:, None] * tt[None, :])
sig += np.where((idx % 3 == 1)[:, None], 0.25 * np.sin(2 * np.pi * (2.5 * f0)[:, None] * tt[None, :]), 0.0)
sig *= np.where((idx % 5 == 2)[:, None], (tt[None, :] > 0.3), 1.0)
sig = (np.round(np.clip(sig, -1, 1 - 2.0 ** -15) * 32768.0) / 32768.0).astype(np.float32)
clean = torch.from_numpy(sig).to(dev)
# a mixed population: the random noise-free signals of tests/test_gpu_parity.py (about 90 % of the clips marked, ~60 % of their frames)
_src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "test_gpu_parity.py")).read()
_ns = {"np": np}
exec(_src[_src.index("def _random_noise_free_signals"):_src.index("def test_logmel_auto_mode_on_2048_random_noise_free_signals")], _ns)
rnd = np.concatenate([_ns["_random_noise_free_signals"](2048, 7)[0], _ns["_random_noise_free_signals"](2048, 8)[0]])
random_clean = torch.from_numpy(rnd).to(dev)
res = {}
ref = mel_oracle.logmel_batch(sig[:48], normalize=True)
for mode in ("f32", "f64", "auto"):
    ops.set_logmel_math(mode)
    for name, x in (("noisy", noisy), ("clean", clean), ("random_clean", random_clean)):
        for _ in range(3):
            out = ops.logmel(x, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = ops.logmel(x, True)
        e1.record()
        torch.cuda.synchronize()
        res[f"{mode}_{name}_ms"] = round(e0.elapsed_time(e1) / 10, 4)
    res[f"{mode}_clean_err_dB"] = float(np.abs(ops.logmel(clean[:48], True).cpu().numpy() - ref).max())
ops.set_logmel_math("auto")
print(json.dumps(res))
