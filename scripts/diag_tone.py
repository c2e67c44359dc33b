import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops
from oracle import mel_oracle
t = np.arange(16000) / 16000.0
x = np.stack([np.sin(2 * np.pi * 440.0 * t), np.sin(2 * np.pi * 3999.5 * t) * 0.01,
              (np.arange(16000) == 8000).astype(np.float64), np.sign(np.sin(2 * np.pi * 100 * t))]).astype(np.float32)
out = ops.logmel(torch.from_numpy(x).cuda(), True).cpu().numpy()
ref = mel_oracle.logmel_batch(x, normalize=True)
for i in range(4):
    e = np.abs(out[i,0]-ref[i,0]); f64 = mel_oracle.logmel_f64(x[i]); e64 = np.abs(ref[i,0]-f64); eo = np.abs(out[i,0]-f64)
    print("clip", i, "max err vs oracle32 %.3e at level %.1f dB | oracle32 vs f64 %.3e | gpu vs f64 %.3e" % (e.max(), ref[i,0].flat[e.argmax()], e64.max(), eo.max()))
    for lo,hi in [(-20,0.1),(-40,-20),(-60,-40),(-80.1,-60)]:
        m = (ref[i,0]>=lo)&(ref[i,0]<hi)
        if m.any(): print("   level [%g,%g): n=%d gpu-vs-oracle %.2e  oracle32-vs-f64 %.2e" % (lo,hi,m.sum(), e[m].max(), e64[m].max()))
