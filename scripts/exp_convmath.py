import numpy as np, torch, time
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops
from oracle import mel_oracle, model_oracle
dev = torch.device("cuda", 0)
g = dict(np.load("tests/golden/model_simple_seed1234.npz"))
sd = pkg.synth.make_state_dict("simple", seed=1234)
packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
x = torch.from_numpy(g["x32"]).to(dev)
big = torch.from_numpy(mel_oracle.logmel_batch(pkg.synth.make_clips(0, 32))).to(dev).repeat(128, 1, 1, 1)
ref_pooled = model_oracle.pooled_features_np(g["x32"], sd); ref_logits = model_oracle.forward_np(g["x32"], sd)
for mode in ("f32", "f16x3"):
    ops.set_conv_math(mode)
    pooled = ops.cnn_pool(x, packed, 2).cpu().numpy(); logits = ops.cnn_lstm_forward(x, packed, 2).cpu().numpy()
    print(mode, "pooled max err vs f64 oracle %.3e (rel %.3e)  logits %.3e | vs ref golden pooled %.3e logits %.3e" % (
        np.abs(pooled-ref_pooled).max(), (np.abs(pooled-ref_pooled)/np.abs(ref_pooled).max()).max(), np.abs(logits-ref_logits).max(),
        np.abs(pooled-g["pooled32"]).max(), np.abs(logits-g["logits32"]).max()))
    for _ in range(3): ops.cnn_pool(big, packed, 2)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): ops.cnn_pool(big, packed, 2)
    torch.cuda.synchronize(); print("   cnn_pool 4096 clips: %.3f ms" % ((time.perf_counter()-t)*100))
