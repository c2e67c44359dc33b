"""CPU-only checks of the native library's host side: it loads, exports exactly what include/wakeword_amd.h
declares, builds the front-end tables and the packed weight image correctly, and refuses to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import wakeword_jupyterlab_amd as pkg
from oracle import mel_oracle
from wakeword_jupyterlab_amd import _native as nat
from wakeword_jupyterlab_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAS_GPU = torch.cuda.is_available()


def _header_functions():
    text = open(os.path.join(ROOT, "include", "wakeword_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"WW_API\s+[\w\s\*]+?\b(ww_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol_and_binding_covers_them():
    names = _header_functions()
    assert len(names) >= 19 and "ww_logmel_f32" in names and "ww_streamer_step" in names
    lib = C.CDLL(nat.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(nat.PROTOTYPES) == names                 # ctypes table == header, nothing more, nothing less
    assert nat.lib.ww_abi_version() == nat.ABI_VERSION == 4
    # nothing else leaks out of the shared object
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", nat.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert [e for e in exported if e.startswith("ww_")] == names


def test_mel_filterbank_and_window_match_the_oracle_bit_for_bit():
    M = np.empty((80, 1025), np.float32)
    nat.check(nat.lib.ww_mel_filterbank_host(M.ctypes.data))
    ref = mel_oracle.mel_filterbank()
    assert (M != 0).sum() == 2004
    assert np.array_equal(M != 0, ref != 0)
    assert np.abs(M - ref).max() <= 2e-9 and np.abs(M / np.where(ref == 0, 1, ref) - (ref != 0)).max() <= 1.2e-7   # <= 1 ulp
    w = np.empty(2048, np.float32)
    nat.check(nat.lib.ww_hann_window_host(w.ctypes.data))
    assert np.abs(w.astype(np.float64) - mel_oracle.hann_window()).max() <= 6e-8 and w[0] == 0.0 and w[1024] == 1.0


@pytest.mark.parametrize("arch", ["simple", "full"])
def test_packed_weight_image_layout(arch):
    sd = pkg.synth.make_state_dict(arch, seed=5)
    n_conv = 2 if arch == "simple" else 3
    p = ops.pack_state_dict(sd)
    assert p.dtype == np.float32 and p.size == nat.lib.ww_packed_weights_floats(n_conv)
    o = 0
    def take(n):
        nonlocal o
        v = p[o:o + n]; o += (n + 3) // 4 * 4
        return v
    assert np.array_equal(take(288), sd["conv1.weight"].reshape(-1))
    assert np.array_equal(take(32), sd["conv1.bias"])
    # conv B operand: [ntile][(c*3+dy)*3+dx][lane] = W[32*nt + lane%32][2c + lane//32][dy][dx]
    for li, cin, cout in [(2, 32, 64), (3, 64, 128)][: n_conv - 1]:
        wb = take(cout // 32 * (cin // 2 * 9) * 64).reshape(cout // 32, cin // 2, 3, 3, 2, 32)
        w = sd[f"conv{li}.weight"]                                    # [cout][cin][3][3]
        want = w.reshape(cout // 32, 32, cin // 2, 2, 3, 3).transpose(0, 2, 4, 5, 3, 1)
        assert np.array_equal(wb, want)
        assert np.array_equal(take(cout), sd[f"conv{li}.bias"])
    # LSTM: [K/16][768][16]; column (hb*3 + g)*32 + u <- row goff[g] + 32*hb + u, gates (i, g, o) = rows 0, 512, 768;
    # inside a group of 16 consecutive k the order is by (k mod 4, k div 4): 0,4,8,12,1,5,9,13,...
    order = np.array([4 * (s % 4) + s // 4 for s in range(16)])
    for layer, K in enumerate([64 if arch == "simple" else 128, 256]):
        wt = take(K * 768).reshape(K // 16, 8, 3, 32, 16)            # [kg][hb][gate][u][slot]
        b = take(768).reshape(8, 3, 32)
        w_ih = sd[f"lstm.weight_ih_l{layer}"]
        bias = sd[f"lstm.bias_ih_l{layer}"] + sd[f"lstm.bias_hh_l{layer}"]
        for g, off in enumerate([0, 512, 768]):
            rows = (off + 32 * np.arange(8)[:, None] + np.arange(32)[None]).reshape(-1)           # [hb*32 + u]
            want = w_ih[rows].reshape(8, 32, K // 16, 16)[:, :, :, order].transpose(2, 0, 1, 3)    # [kg][hb][u][slot]
            assert np.array_equal(wt[:, :, g], want)
            assert np.array_equal(b[:, g, :].reshape(-1), bias[rows])
    assert np.array_equal(take(512), sd["fc.weight"].reshape(-1))
    assert np.array_equal(take(4)[:2], sd["fc.bias"])
    # split-precision images: every OUTPUT CHANNEL (row) carries its own power-of-two scale, w * 2^S[row] = hi + lo (two f16
    # halves) with the row's largest |w'| in [2^12, 2^13): good to 2^-21 of the ROW's largest weight
    def row_exps(descale, wrows):
        S = -np.round(np.log2(descale.astype(np.float64))).astype(int)
        assert np.array_equal(descale, (2.0 ** -S).astype(np.float32))                 # exact powers of two
        m = np.abs(wrows).max(axis=1) * 2.0 ** S
        assert np.all((m >= 2.0 ** 12) & (m < 2.0 ** 13))
        return S
    w2 = sd["conv2.weight"].astype(np.float64)
    S2 = row_exps(take(64), w2.reshape(64, -1))
    if n_conv == 3:
        w3 = sd["conv3.weight"].astype(np.float64)
        h3 = take(8 * 18 * 2 * 64 * 4).view(np.float16).astype(np.float64).reshape(8, 18, 2, 64, 8)   # [nt16][ks][hi/lo][lane][j]
        S3 = row_exps(take(128), w3.reshape(128, -1))
        nt3, cb3, dx3, dy3, lane3, j3 = np.meshgrid(np.arange(8), np.arange(2), np.arange(3), np.arange(3), np.arange(64), np.arange(8), indexing="ij")
        co3 = 16 * nt3 + (lane3 & 15)
        want3 = w3[co3, 32 * cb3 + 8 * (lane3 >> 4) + j3, dy3, dx3] * 2.0 ** S3[co3]                   # ks = (cb*3 + dx)*3 + dy
        assert np.abs((h3[:, :, 0] + h3[:, :, 1]).reshape(8, 2, 3, 3, 64, 8) - want3).max() <= 2.0 ** 13 * 2.0 ** -21
    c1 = take(2 * 64 * 4).view(np.float16).astype(np.float64).reshape(2, 64, 8)                      # conv1 A operand: 9 taps, 7 zeros
    hs1 = take(4)
    S1 = int(row_exps(hs1[:1], sd["conv1.weight"].astype(np.float64).reshape(1, -1))[0])             # one scale for the tensor
    lane, j = np.meshgrid(np.arange(64), np.arange(8), indexing="ij")
    k = 8 * (lane >> 5) + j
    w1z = np.concatenate([sd["conv1.weight"].reshape(32, 9), np.zeros((32, 7), np.float32)], 1).astype(np.float64) * 2.0 ** S1
    mrow = lane & 31
    assert np.abs((c1[0] + c1[1]) - w1z[(mrow & 3) + 4 * (mrow >> 3) + 16 * ((mrow >> 2) & 1), k]).max() <= 2.0 ** 13 * 2.0 ** -21
    h16 = take(4 * 9 * 2 * 64 * 4).view(np.float16).astype(np.float64).reshape(4, 9, 2, 64, 8)       # [nt16][ks = dx*3+dy][hi/lo][lane][j]
    nt, dx, dy, lane, j = np.meshgrid(np.arange(4), np.arange(3), np.arange(3), np.arange(64), np.arange(8), indexing="ij")
    co = 16 * nt + (lane & 15)
    want16 = w2[co, 8 * (lane >> 4) + j, dy, dx] * 2.0 ** S2[co]
    assert np.abs((h16[:, :, 0] + h16[:, :, 1]).reshape(4, 3, 3, 64, 8) - want16).max() <= 2.0 ** 13 * 2.0 ** -21
    # LSTM W_ih split for the 16x16x32 f16 MFMA: [K/32][48 ntile][hi/lo][64 lanes][8]; column c = 16 nt + lane%16 in the
    # same (hb, gate, u) order as the f32 image, k = 32 kb + 8 (lane//16) + j; one scale per packed column
    Ks = [64 if arch == "simple" else 128, 256]
    lh = [take(K // 32 * 48 * 2 * 64 * 4).view(np.float16).astype(np.float64).reshape(K // 32, 48, 2, 64, 8) for K in Ks]
    hs = take(2 * 768).reshape(2, 768)
    goff = np.array([0, 512, 768])
    cols = np.arange(768)
    col_row = goff[(cols % 96) // 32] + 32 * (cols // 96) + cols % 32
    for layer, K in enumerate(Ks):
        w_ih = sd[f"lstm.weight_ih_l{layer}"].astype(np.float64)
        S = row_exps(hs[layer], w_ih[col_row])
        kb, nt, lane, j = np.meshgrid(np.arange(K // 32), np.arange(48), np.arange(64), np.arange(8), indexing="ij")
        c = 16 * nt + (lane & 15)
        want = w_ih[col_row[c], 32 * kb + 8 * (lane >> 4) + j] * 2.0 ** S[c]
        assert np.abs((lh[layer][:, :, 0] + lh[layer][:, :, 1]) - want).max() <= 2.0 ** 13 * 2.0 ** -21
    # conv2 in 1-D Winograd F(2,3) form along the rows: U0 = w[dy=0], U1 = (w0+w1+w2)/2, U2 = (w0-w1+w2)/2, U3 = w[dy=2];
    # [nt16][ks = xi*3 + dx][hi/lo][lane][j], one scale per output channel over all four xi
    hw = take(4 * 12 * 2 * 64 * 4).view(np.float16).astype(np.float64).reshape(4, 4, 3, 2, 64, 8)      # [nt][xi][dx][hi/lo][lane][j]
    U = np.stack([w2[:, :, 0, :], 0.5 * (w2[:, :, 0, :] + w2[:, :, 1, :] + w2[:, :, 2, :]),
                  0.5 * (w2[:, :, 0, :] - w2[:, :, 1, :] + w2[:, :, 2, :]), w2[:, :, 2, :]], axis=2)     # [co][ci][xi][dx]
    Sw = row_exps(take(64), U.reshape(64, -1))
    nt, xi, dx, lane, j = np.meshgrid(np.arange(4), np.arange(4), np.arange(3), np.arange(64), np.arange(8), indexing="ij")
    co = 16 * nt + (lane & 15)
    wantw = U[co, 8 * (lane >> 4) + j, xi, dx] * 2.0 ** Sw[co]
    assert np.abs((hw[:, :, :, 0] + hw[:, :, :, 1]) - wantw).max() <= 2.0 ** 13 * 2.0 ** -21
    if n_conv == 3:       # conv3 in the same form: 2 channel blocks, ks = (xi*3 + dx)*2 + cb
        hw3 = take(8 * 24 * 2 * 64 * 4).view(np.float16).astype(np.float64).reshape(8, 4, 3, 2, 2, 64, 8)   # [nt][xi][dx][cb][hi/lo][lane][j]
        U3 = np.stack([w3[:, :, 0, :], 0.5 * (w3[:, :, 0, :] + w3[:, :, 1, :] + w3[:, :, 2, :]),
                       0.5 * (w3[:, :, 0, :] - w3[:, :, 1, :] + w3[:, :, 2, :]), w3[:, :, 2, :]], axis=2)
        Sw3 = row_exps(take(128), U3.reshape(128, -1))
        nt, xi, dx, cb, lane, j = np.meshgrid(np.arange(8), np.arange(4), np.arange(3), np.arange(2), np.arange(64), np.arange(8), indexing="ij")
        co = 16 * nt + (lane & 15)
        want = U3[co, 32 * cb + 8 * (lane >> 4) + j, xi, dx] * 2.0 ** Sw3[co]
        assert np.abs((hw3[:, :, :, :, 0] + hw3[:, :, :, :, 1]) - want).max() <= 2.0 ** 13 * 2.0 ** -21
    # range bounds for the per-clip activation exponents: |conv_l out| <= max|in| * l1[l] + max|b_l|
    rng = take(8)
    for li, (l1, bm) in enumerate([(rng[0], rng[1]), (rng[2], rng[3])], start=1):
        w = sd[f"conv{li}.weight"].astype(np.float64)
        true_l1 = np.abs(w).reshape(w.shape[0], -1).sum(axis=1).max()
        assert true_l1 <= l1 <= true_l1 * 1.0001 and bm == np.abs(sd[f"conv{li}.bias"]).max()
    # conv2's Winograd form again as B operands of v_mfma_f32_32x32x16_f16 (cnn2x_kernel): [nt32][xi][step = dx*2 + c][hi/lo][lane][j], lane
    # (n = lane % 32, hh = lane // 32) holds U_xi[dx][ci = 16 c + 8 hh + j][co = 32 nt + n]; the scales are the 16x16x32 image's (Sw)
    hx = take(2 * 4 * 6 * 2 * 64 * 4).view(np.float16).astype(np.float64).reshape(2, 4, 3, 2, 2, 64, 8)     # [nt][xi][dx][c][hi/lo][lane][j]
    nt, xi, dx, c, lane, j = np.meshgrid(np.arange(2), np.arange(4), np.arange(3), np.arange(2), np.arange(64), np.arange(8), indexing="ij")
    co = 32 * nt + (lane & 31)
    wantx = U[co, 16 * c + 8 * (lane >> 5) + j, xi, dx] * 2.0 ** Sw[co]
    assert np.abs((hx[:, :, :, :, 0] + hx[:, :, :, :, 1]) - wantx).max() <= 2.0 ** 13 * 2.0 ** -21
    assert o == p.size


def test_pack_rejects_bad_state_dicts():
    sd = pkg.synth.make_state_dict("simple")
    bad = dict(sd); del bad["fc.bias"]
    with pytest.raises(KeyError):
        ops.pack_state_dict(bad)
    bad = dict(sd); bad["conv2.weight"] = np.zeros((64, 32, 5, 5), np.float32)
    with pytest.raises(ValueError):
        ops.pack_state_dict(bad)
    s = nat.StateDict(); s.n_conv = 4; s.hidden = 256
    out = np.zeros(16, np.float32)
    assert nat.lib.ww_pack_weights_host(C.byref(s), out.ctypes.data) == nat.WW_EINVAL
    assert b"n_conv" in nat.lib.ww_last_error()
    assert nat.lib.ww_packed_weights_floats(5) == nat.WW_EINVAL


def test_resampler_taps_match_scipy_design():
    from scipy.signal import firwin
    for sr, (u, d) in {44100: (160, 441), 48000: (1, 3), 8000: (2, 1), 22050: (320, 441), 32000: (1, 2)}.items():
        up, dn, hl = C.c_int32(), C.c_int32(), C.c_int32()
        n = nat.lib.ww_resample_taps_host(sr, None, 0, C.byref(up), C.byref(dn), C.byref(hl))
        assert (up.value, dn.value, hl.value, n) == (u, d, 10 * max(u, d), 20 * max(u, d) + 1)
        taps = np.zeros(n, np.float32)
        assert nat.lib.ww_resample_taps_host(sr, taps.ctypes.data, n, None, None, None) == n
        ref = firwin(n, 1.0 / max(u, d), window=("kaiser", 5.0)) * u     # scipy.signal.resample_poly's filter
        assert np.abs(taps - ref).max() <= 6e-8 * max(1, u)
    assert nat.lib.ww_resample_taps_host(16000, None, 0, None, None, None) == 0        # already 16 kHz: no filter
    assert nat.lib.ww_resample_taps_host(10, None, 0, None, None, None) == nat.WW_EINVAL


def test_size_queries_do_not_need_a_gpu():
    assert nat.lib.ww_workspace_bytes(4096, 2) >= 4096 * (2560 + 64) * 4
    # relu(conv2) records of the 3-conv model + one per-clip scale float (256-byte aligned block)
    assert nat.lib.ww_cnn_scratch_bytes(16, 2) == 0 and nat.lib.ww_cnn_scratch_bytes(16, 3) == 16 * 80 * 64 * 32 * 4 + 256


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_compute_entry_points_fail_loudly_without_a_gpu():
    # no CPU fallback anywhere: launches return WW_ENODEVICE, torch ops on CPU tensors raise
    buf = np.zeros(16000, np.float32)
    out = np.zeros(2560, np.float32)
    rc = nat.lib.ww_logmel_f32(buf.ctypes.data, 1, 16000, 16000, 1, out.ctypes.data, None)
    assert rc == nat.WW_ENODEVICE and b"no HIP device" in nat.lib.ww_last_error()
    assert nat.lib.ww_init() == nat.WW_ENODEVICE
    with pytest.raises(RuntimeError):
        ops.logmel(torch.zeros(1, 16000), True)
    with pytest.raises(RuntimeError):
        torch.ops.wakeword_amd.forward_pcm(torch.zeros(1, 16000), torch.zeros(4), 2, True)
    m = pkg.SimpleWakewordModel().eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 80, 32))
    with pytest.raises(RuntimeError):
        pkg.AudioProcessor().audio_to_mel(np.ones(16000, np.float32))


def test_training_workspace_size_is_checked_against_the_mode_of_the_call():
    """the workspace layout follows the arithmetic (0.23 GB vs 2.7 GB at 4096 clips), so the mode and the buffer size are
    arguments of every call of a step; a buffer sized for the other mode is WW_EINVAL before anything is launched (no GPU needed)."""
    small, large = nat.lib.ww_train_workspace_bytes(4096, 2, 1), nat.lib.ww_train_workspace_bytes(4096, 2, 0)
    assert 0 < small < 0.3e9 < 2e9 < large
    assert nat.lib.ww_train_workspace_bytes(4096, 2, -1) == small                 # default mode of a fresh process: f16x3
    assert nat.lib.ww_train_workspace_bytes(4096, 2, 7) == nat.WW_EINVAL
    fake = np.zeros(1024, np.float32)
    base = (fake.ctypes.data + 255) // 256 * 256
    tp = nat.TrainParams()
    tp.n_conv, tp.hidden = 2, 256
    for i in range(2):
        tp.conv_weight[i] = tp.conv_bias[i] = tp.lstm_weight_ih[i] = tp.lstm_bias_ih[i] = tp.lstm_bias_hh[i] = base
    tp.fc_weight = tp.fc_bias = base
    args = (base, 4096, 32, C.byref(tp), 0.0, 0.0, 1)
    assert nat.lib.ww_train_forward_f32(*args, 0, base, small, base, None) == nat.WW_EINVAL          # f32 layout into an f16x3-sized buffer
    assert b"need" in nat.lib.ww_last_error()
    assert nat.lib.ww_train_forward_f32(*args, 5, base, large, base, None) == nat.WW_EINVAL          # unknown mode
    tg = nat.TrainGrads()
    assert nat.lib.ww_train_backward_f32(base, 4096, 32, C.byref(tp), base, 0, base, small, C.byref(tg), None) == nat.WW_EINVAL
    if not HAS_GPU:                                                                                   # large enough: only the device is missing
        assert nat.lib.ww_train_forward_f32(*args, 1, base, small, base, None) == nat.WW_ENODEVICE


def test_argument_checks_come_before_device_checks():
    buf = np.zeros(16, np.float32)
    assert nat.lib.ww_logmel_f32(buf.ctypes.data, 1, 16000, 20000, 1, buf.ctypes.data, None) == nat.WW_EINVAL
    assert nat.lib.ww_logmel_f32(buf.ctypes.data + 4, 2, 16000, 16000, 1, buf.ctypes.data, None) == nat.WW_EINVAL
    assert nat.lib.ww_cnn_pool_f32(buf.ctypes.data, 1, 40, buf.ctypes.data, 2, None, buf.ctypes.data, None) == nat.WW_EUNSUPPORTED
    assert nat.lib.ww_lstm_fc_f32(buf.ctypes.data, 1, buf.ctypes.data, 7, buf.ctypes.data, None) == nat.WW_EINVAL


def test_header_is_plain_c_and_library_links_from_c(tmp_path):
    """include/wakeword_amd.h compiles as strict C99 and a C program drives the library (tests/c/abi_host_check.c)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "wakeword-jupyterlab_amd")
    exe = os.path.join(tmp_path, "abi_host_check")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c", "abi_host_check.c"), "-o", exe, "-L", libdir, "-l:libwakeword_amd.so", "-lm",
                    "-Wl,-rpath," + libdir], check=True, capture_output=True, text=True)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.dirname(torch.__file__) + "/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, "gpu" if HAS_GPU else "cpu", os.path.join(tmp_path, "abi_check.wav")], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "abi_host_check OK" in r.stdout, r.stderr


def test_custom_ops_are_compiled_operators_with_meta_kernels():
    """torch.ops.wakeword_amd.* come from libwakeword_amd_torch.so (TORCH_LIBRARY + CUDA / Meta / CPU kernels),
    not from Python registrations: the schemas are there, meta tensors and FakeTensors (what torch.compile / export trace with) get their
    output shapes without touching a device, CPU tensors are refused."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from wakeword_jupyterlab_amd import ops
    assert os.path.exists(ops.TORCH_LIB_PATH)
    loaded = open("/proc/self/maps").read()
    assert "libwakeword_amd_torch.so" in loaded and "libwakeword_amd.so" in loaded
    ns = torch.ops.wakeword_amd
    assert str(ns.logmel.default._schema) == "wakeword_amd::logmel(Tensor pcm, bool normalize=True) -> Tensor"
    assert str(ns.forward_pcm.default._schema) == "wakeword_amd::forward_pcm(Tensor pcm, Tensor packed, int n_conv, bool normalize=True) -> Tensor"
    m = lambda *s: torch.empty(*s, device="meta")                               # noqa: E731
    assert ns.logmel(m(4, 16000), True).shape == (4, 1, 80, 32) and ns.logmel(m(0, 777)).shape == (0, 1, 80, 32)
    assert ns.cnn_pool(m(3, 1, 80, 31), m(9), 2).shape == (3, 64) and ns.cnn_pool(m(3, 1, 80, 32), m(9), 3).shape == (3, 128)
    assert ns.lstm_fc(m(5, 128), m(9), 3).shape == (5, 2)
    assert ns.cnn_lstm_forward(m(6, 1, 80, 32), m(9), 2).shape == (6, 2)
    assert ns.forward_pcm(m(7, 9000), m(9), 3, False).shape == (7, 2)
    for bad in (lambda: ns.logmel(m(4, 16001)), lambda: ns.cnn_pool(m(3, 1, 80, 33), m(9), 2), lambda: ns.lstm_fc(m(5, 64), m(9), 3),
                lambda: ns.forward_pcm(m(7, 9000), m(9), 4)):
        with pytest.raises(RuntimeError):
            bad()
    with FakeTensorMode():                                                       # fake CUDA tensors on a machine without a GPU
        y = ns.cnn_lstm_forward(torch.empty(5, 1, 80, 32, device="cuda"), torch.empty(100, device="cuda"), 2)
        z = ops.logmel(torch.empty(2, 16000, device="cuda"), True)               # the Python wrapper's checks run on fake tensors too
        assert y.shape == (5, 2) and y.device.type == "cuda" and z.shape == (2, 1, 80, 32)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        ns.lstm_fc(torch.zeros(2, 64), torch.zeros(4), 2)
