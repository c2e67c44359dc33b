"""A/B timing of one C-ABI entry point across several builds of the library, interleaved in one process
(cdna_hip_programming.md rule 24).  usage: ab_kernels.py logmel|cnn|cnn3|head|full libA.so libB.so ...
Every build packs its own weight image (the packed layout may differ between builds) and must reproduce the first build's
result to 1e-4 (pooled) / bitwise (logmel)."""
import ctypes as C, sys, time
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat
from wakeword_jupyterlab_amd import ops

what, libs = sys.argv[1], sys.argv[2:]
dev = torch.device("cuda", 0)
B = 4096
pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, B, unique=64)).to(dev)
mel = ops.logmel(pcm, True)
out = torch.empty_like(mel); lg = torch.empty(B, 2, device=dev)
arch = "full" if what in ("cnn3", "full3") else "simple"
n_conv = 3 if arch == "full" else 2
sd = pkg.synth.make_state_dict(arch)


def pack_with(h):
    h.ww_packed_weights_floats.restype = C.c_int64
    chans = [1, 32, 64, 128][: n_conv + 1]
    keep, s = [], nat.StateDict()
    s.n_conv, s.hidden = n_conv, 256
    def arr(k):
        v = np.ascontiguousarray(sd[k], np.float32); keep.append(v); return v.ctypes.data
    for i in range(n_conv):
        s.conv_weight[i], s.conv_bias[i] = arr(f"conv{i+1}.weight"), arr(f"conv{i+1}.bias")
    for l in range(2):
        s.lstm_weight_ih[l], s.lstm_bias_ih[l], s.lstm_bias_hh[l] = arr(f"lstm.weight_ih_l{l}"), arr(f"lstm.bias_ih_l{l}"), arr(f"lstm.bias_hh_l{l}")
    s.fc_weight, s.fc_bias = arr("fc.weight"), arr("fc.bias")
    img = np.empty(h.ww_packed_weights_floats(C.c_int32(n_conv)), np.float32)
    assert h.ww_pack_weights_host(C.byref(s), C.c_void_p(img.ctypes.data)) == 0
    return torch.from_numpy(img).to(dev)


hs, packs = [], []
for path in libs:
    h = C.CDLL(path); h.ww_init(); hs.append(h); packs.append(pack_with(h))
    import os
    if os.environ.get("AB_LOGMEL_MATH") and hasattr(h, "ww_set_logmel_math"):      # 0 f32, 1 f64, 2 auto
        assert h.ww_set_logmel_math(int(os.environ["AB_LOGMEL_MATH"])) == 0
    h.ww_cnn_scratch_bytes.restype = C.c_int64; h.ww_workspace_bytes.restype = C.c_int64
scratch = torch.empty(max(1, max(h.ww_cnn_scratch_bytes(C.c_int64(B), C.c_int32(n_conv)) for h in hs)), dtype=torch.uint8, device=dev)
ws = torch.empty(max(h.ww_workspace_bytes(C.c_int64(B), C.c_int32(n_conv)) for h in hs), dtype=torch.uint8, device=dev)
pooled = torch.empty(B, 64 * (n_conv - 1), device=dev)


def run(h, pk):
    if what == "logmel":
        rc = h.ww_logmel_f32(C.c_void_p(pcm.data_ptr()), C.c_int64(B), C.c_int64(16000), C.c_int64(16000), 1, C.c_void_p(out.data_ptr()), None)
    elif what == "head":
        rc = h.ww_lstm_fc_f32(C.c_void_p(pooled.data_ptr()), C.c_int64(B), C.c_void_p(pk.data_ptr()), n_conv, C.c_void_p(lg.data_ptr()), None)
    elif what in ("full", "full3"):     # PCM -> logits
        rc = h.ww_forward_pcm_f32(C.c_void_p(pcm.data_ptr()), C.c_int64(B), C.c_int64(16000), C.c_int64(16000), 1, C.c_void_p(pk.data_ptr()),
                                  n_conv, C.c_void_p(ws.data_ptr()), C.c_void_p(lg.data_ptr()), None)
    else:
        rc = h.ww_cnn_pool_f32(C.c_void_p(mel.data_ptr()), C.c_int64(B), 32, C.c_void_p(pk.data_ptr()), n_conv,
                               C.c_void_p(scratch.data_ptr()), C.c_void_p(pooled.data_ptr()), None)
    assert rc == 0, rc


res = {p: [] for p in libs}
first = None
for path, h, pk in zip(libs, hs, packs):
    for _ in range(3): run(h, pk)
    torch.cuda.synchronize()
    got = {"logmel": out, "head": lg, "full": lg, "full3": lg}.get(what, pooled).clone()
    if first is None:
        first = got
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run(h, pk)
    torch.cuda.synchronize()
    print("%-40s max |diff| vs first build %.3g   (first look: %.4f ms)" % (path.split("/")[-1], float((got - first).abs().max()),
                                                                           (time.perf_counter() - t0) * 200), flush=True)
for rnd in range(12):
    for p, h, pk in zip(libs, hs, packs):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): run(h, pk)
        torch.cuda.synchronize(); res[p].append((time.perf_counter() - t) * 100)
for p in libs:
    v = np.array(res[p]); print("%-40s median %.4f ms  min %.4f ms" % (p.split("/")[-1], np.median(v), v.min()))
