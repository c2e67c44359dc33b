"""Phase stamps of the conv kernel (diagnostic build with -DWW_STAMPS): share of wave time per phase for consumer wave 1 and producer wave 9
of workgroup 7.  usage: PYTHONPATH=. python scripts/k2_stamps.py lib.so [launches]"""
import ctypes as C, sys
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat, ops

dev = torch.device("cuda", 0)
B = 4096
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev)
mel = ops.logmel(pcm, True)
sd = pkg.synth.make_state_dict("simple")
pooled = torch.empty(B, 64, device=dev)
h = C.CDLL(sys.argv[1]); h.ww_init()
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 50
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
for _ in range(20): run()
torch.cuda.synchronize()
z = (C.c_ulonglong * 16)()
h.ww_debug_cnn_stamps(z)
for _ in range(launches): run()
torch.cuda.synchronize()
h.ww_debug_cnn_stamps(z)
v = np.array(list(z), dtype=np.float64) / launches
tiles_c, tiles_p = 16 * 20, 16 * 10
print("consumer wave 1: cycles per tile row by phase", np.round(v[:8] / tiles_c, 1), "total", round(v[:8].sum() / tiles_c, 1))
print("producer wave 9: cycles per tile row by phase", np.round(v[8:] / tiles_p, 1), "total", round(v[8:].sum() / tiles_p, 1))
