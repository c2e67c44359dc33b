"""In-kernel clock of the conv kernel (diagnostic builds with -DWW_CLOCK): >= 2 s of back-to-back launches on random-like data, then the
sums of s_memtime / s_memrealtime deltas of one consumer wave per workgroup.  usage: PYTHONPATH=. python scripts/k2_clock.py lib.so ..."""
import ctypes as C, sys, time
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat, ops

dev = torch.device("cuda", 0)
B = 4096
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev)
mel = ops.logmel(pcm, True)
sd = pkg.synth.make_state_dict("simple")
pooled = torch.empty(B, 64, device=dev)
for path in sys.argv[1:]:
    h = C.CDLL(path); h.ww_init()
    h.ww_packed_weights_floats.restype = C.c_int64
    keep, s = [], nat.StateDict()
    s.n_conv, s.hidden = 2, 256
    def arr(k):
        v = np.ascontiguousarray(sd[k], np.float32); keep.append(v); return v.ctypes.data
    for i in range(2):
        s.conv_weight[i], s.conv_bias[i] = arr(f"conv{i+1}.weight"), arr(f"conv{i+1}.bias")
    for l in range(2):
        s.lstm_weight_ih[l], s.lstm_bias_ih[l], s.lstm_bias_hh[l] = arr(f"lstm.weight_ih_l{l}"), arr(f"lstm.bias_ih_l{l}"), arr(f"lstm.bias_hh_l{l}")
    s.fc_weight, s.fc_bias = arr("fc.weight"), arr("fc.bias")
    img = np.empty(h.ww_packed_weights_floats(C.c_int32(2)), np.float32)
    assert h.ww_pack_weights_host(C.byref(s), C.c_void_p(img.ctypes.data)) == 0
    pk = torch.from_numpy(img).to(dev)
    def run():
        assert h.ww_cnn_pool_f32(C.c_void_p(mel.data_ptr()), C.c_int64(B), 32, C.c_void_p(pk.data_ptr()), 2, None, C.c_void_p(pooled.data_ptr()), None) == 0
    t_end = time.perf_counter() + 2.0
    while time.perf_counter() < t_end:
        for _ in range(50): run()
        torch.cuda.synchronize()
    z = (C.c_ulonglong * 4)()
    h.ww_debug_clock(z)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 5
    h.ww_debug_clock(z)
    print("%-28s %.4f ms/launch  in-kernel clock %.3f GHz  (loop %.1f us per workgroup)" % (path.split("/")[-1], ms, z[0] / z[1] * 0.1, z[1] / z[2] / 100.0), flush=True)
