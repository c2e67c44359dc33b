"""The split-precision (f16x3) default on UNFRIENDLY numbers, and the guards behind it.

Every model test elsewhere draws weights from U(+-1/sqrt(fan_in)) and inputs from the log-mel range.  Here: heavy-tailed
and outlier weights, tensors scaled far up and down, conv1 weights spanning 22 octaves, inputs far outside [-80, 0] dB
(the reference's forward takes any float tensor, /root/reference/wakeword_training/train_wakeword.py:38-49), and
cases whose conv1 activations / pooled features exceed the f16 range (65504) -- which the kernels absorb with
power-of-two exponents per output channel (weights), per clip (inputs, activations) and per row (LSTM inputs).
Oracle: oracle/model_oracle.forward_np (float64 arithmetic on the float32 parameters).  Tolerance: 1e-3 on logits
(north_star); the worst error of every case is written to gpurun_out/guard_errors.json when that directory exists.

The logits alone cannot see the conv arithmetic in most of these cases (saturated gates: a 1e-3 relative change of the
pooled features moves the logits by < 1e-4 in 12 of the 14 weight cases), so every case ALSO compares the conv stack's
output itself: `ops.cnn_pool` against `model_oracle.pooled_features_np`, per clip relative to the clip's largest pooled
feature, under the arithmetic being tested AND under the exact-fp32 kernels on the same inputs.  Required:
    err(split) <= 2 * err(f32 kernels) + 2^-22      and      err(split) <= 1e-5
-- a single f16 MFMA per product block (2^-11) misses both by three orders of magnitude.  The head has its own twin on
rows whose gates are NOT saturated, with a self-check that the case would fail under plain-f16 inputs.

Also: an expired in-kernel wait poisons the workgroup's outputs with NaN (observed through the diagnostic twin
library whose waits expire at once), and single clips of any length go through the C ABI.
"""
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest
import torch

import wakeword_jupyterlab_amd as pkg
from oracle import mel_oracle, model_oracle

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3
POOLED_REL_CAP = 1e-5            # |pooled - float64 oracle| / max|pooled of the clip|, any arithmetic
SPLIT_EPS = 2.0 ** -22           # what the dropped lo*lo term of the split may add on top of the fp32 kernels' own error
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_WORST = {}
_WORST_POOLED = {}
_WORST_HEAD = {}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


@pytest.fixture(scope="module", autouse=True)
def _report():
    yield
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out) and _WORST:
        with open(os.path.join(out, "guard_errors.json"), "w") as f:
            json.dump({"tolerance": LOGIT_TOL, "worst_logit_abs_err": _WORST, "max": max(_WORST.values()),
                       "pooled_rel_cap": POOLED_REL_CAP, "split_eps": SPLIT_EPS,
                       "worst_pooled_rel_err": _WORST_POOLED,      # {case: {"err": this arithmetic, "err_f32_kernels": ...}}
                       "max_pooled_rel_err": max([v["err"] for v in _WORST_POOLED.values()] or [0.0]),
                       "head_unsaturated": _WORST_HEAD}, f, indent=1, sort_keys=True)


@pytest.fixture(params=["f16x3", "f16x3d", "f32"])
def conv_math(request):
    from wakeword_jupyterlab_amd import ops
    ops.set_conv_math(request.param)
    yield request.param
    ops.set_conv_math("f16x3")


def _rng(seed):
    return np.random.default_rng(seed)


def _sd(arch, seed=21):
    return {k: v.copy() for k, v in pkg.synth.make_state_dict(arch, seed=seed).items()}


def _weights_case(arch, case):
    sd = _sd(arch)
    r = _rng(zlib.crc32(case.encode()) % 1000)
    conv_keys = [k for k in sd if k.startswith("conv") and k.endswith("weight")]
    mat_keys = conv_keys + ["lstm.weight_ih_l0", "lstm.weight_ih_l1"]
    if case == "lognormal":                       # heavy tails: |w| * exp(N(0, 2^2)) -> ~23 octaves inside a tensor
        for k in mat_keys:
            sd[k] = (sd[k] * np.exp(2.0 * r.standard_normal(sd[k].shape))).astype(np.float32)
    elif case == "outlier_x1000":                 # one weight >> the rest in every tensor (a per-tensor scale's worst case)
        for k in mat_keys:
            flat = sd[k].reshape(-1)
            flat[int(r.integers(flat.size))] *= 1000.0
    elif case == "scaled_1e-6":
        for k in mat_keys:
            sd[k] = (sd[k] * 1e-6).astype(np.float32)
    elif case == "scaled_1e+3":
        for k in mat_keys:
            sd[k] = (sd[k] * 1e3).astype(np.float32)
    elif case == "conv1_22_octaves":              # conv1 weights +-2^u, u ~ U(-20, 2); biases to match
        sd["conv1.weight"] = (np.sign(r.standard_normal((32, 1, 3, 3))) * 2.0 ** r.uniform(-20, 2, (32, 1, 3, 3))).astype(np.float32)
        sd["conv1.bias"] = (2.0 ** r.uniform(-20, 2, 32) * np.sign(r.standard_normal(32))).astype(np.float32)
    elif case == "tiny_channels":                 # whole output channels 2^-18 below the rest (per-channel scales matter)
        for k in conv_keys[1:] + ["lstm.weight_ih_l0"]:
            sd[k][::3] *= np.float32(2.0 ** -18)
    elif case == "big_bias":
        for k in list(sd):
            if k.startswith("conv") and k.endswith("bias"):
                sd[k] = (sd[k] * 3000.0).astype(np.float32)
    else:
        raise AssertionError(case)
    return sd


def _inputs_case(case, width=32, batch=6):
    r = _rng(zlib.crc32(case.encode()) % 997 + 5)
    if case == "logmel":
        return (r.standard_normal((batch, 1, 80, width)) * 15 - 35).clip(-80, 0).astype(np.float32)
    if case == "randn_x1000":
        return (r.standard_normal((batch, 1, 80, width)) * 1000).astype(np.float32)
    if case == "const_-80":
        return np.full((batch, 1, 80, width), -80.0, np.float32)
    if case == "const_0":
        return np.zeros((batch, 1, 80, width), np.float32)
    if case == "tiny_1e-30":
        return (r.standard_normal((batch, 1, 80, width)) * 1e-30).astype(np.float32)
    if case == "one_hot_1e6":                     # one enormous pixel in an otherwise quiet image
        x = (r.standard_normal((batch, 1, 80, width)) * 1e-2).astype(np.float32)
        x[:, 0, 40, width // 2] = 1e6
        return x
    if case == "mixed_per_clip":                  # every clip its own magnitude: the exponents are per clip
        x = r.standard_normal((batch, 1, 80, width)).astype(np.float32)
        return (x * (10.0 ** np.arange(-4, -4 + 2 * batch, 2))[:, None, None, None]).astype(np.float32)
    raise AssertionError(case)


def _run(arch, sd, x, dev):
    m = pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    with torch.no_grad():
        return m(torch.from_numpy(x).to(dev)).cpu().numpy()


def _pooled_rel_err(arch, sd, x, dev, ref):
    """max over clips of max_c |cnn_pool - ref| / max_c |ref| (the clip's largest pooled feature), current conv math."""
    from wakeword_jupyterlab_amd import ops
    n_conv = 2 if arch == "simple" else 3
    packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
    got = ops.cnn_pool(torch.from_numpy(x).to(dev), packed, n_conv).cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all(), "non-finite pooled features where the reference is finite"
    scale = np.abs(ref).max(axis=1, keepdims=True)
    dead = scale[:, 0] == 0.0
    assert (got[dead] == 0.0).all(), "a clip whose pooled features are all exactly 0 in float64 must come out 0"
    return float((np.abs(got - ref)[~dead] / scale[~dead]).max()) if (~dead).any() else 0.0


def _check(tag, arch, sd, x, dev, conv_math):
    from wakeword_jupyterlab_amd import ops
    ref = model_oracle.forward_np(x, sd)
    assert np.isfinite(ref).all(), "the float64 oracle itself is not finite: bad test case"
    y = _run(arch, sd, x, dev)
    assert np.isfinite(y).all(), f"{tag}: non-finite logits where the reference is finite"
    err = float(np.abs(y - ref).max())
    _WORST[f"{arch}/{conv_math}/{tag}"] = err
    assert err <= LOGIT_TOL, f"{tag}: logits max |err| = {err:.3e}"
    # the conv stack itself, where the logits cannot hide it
    pref = model_oracle.pooled_features_np(x, sd)
    perr = _pooled_rel_err(arch, sd, x, dev, pref)
    if conv_math == "f32":
        perr32 = perr
    else:
        ops.set_conv_math("f32")
        try:
            perr32 = _pooled_rel_err(arch, sd, x, dev, pref)
        finally:
            ops.set_conv_math(conv_math)
    _WORST_POOLED[f"{arch}/{conv_math}/{tag}"] = {"err": perr, "err_f32_kernels": perr32}
    assert perr <= POOLED_REL_CAP, f"{tag}: pooled features off by {perr:.3e} of the clip's largest (cap {POOLED_REL_CAP:.0e})"
    assert perr <= 2 * perr32 + SPLIT_EPS, f"{tag}: pooled rel. err {perr:.3e} under {conv_math} vs {perr32:.3e} under the exact-fp32 kernels"


@pytest.mark.parametrize("arch", ["simple", "full"])
@pytest.mark.parametrize("case", ["lognormal", "outlier_x1000", "scaled_1e-6", "scaled_1e+3", "conv1_22_octaves",
                                  "tiny_channels", "big_bias"])
def test_adversarial_weights(dev, conv_math, arch, case):
    _check("w:" + case, arch, _weights_case(arch, case), _inputs_case("logmel"), dev, conv_math)


@pytest.mark.parametrize("arch", ["simple", "full"])
@pytest.mark.parametrize("case", ["randn_x1000", "const_-80", "const_0", "tiny_1e-30", "one_hot_1e6", "mixed_per_clip"])
def test_inputs_outside_the_logmel_range(dev, conv_math, arch, case):
    _check("x:" + case, arch, _sd(arch), _inputs_case(case, width=31 if case == "mixed_per_clip" else 32), dev, conv_math)


@pytest.mark.parametrize("arch", ["simple", "full"])
def test_f16_range_guard_activations_beyond_65504(dev, conv_math, arch):
    """conv1 weights x 8000 on log-mel inputs: relu(conv1) reaches ~8e5, relu(conv2) ~1e6 and the pooled features ~1e5 --
    all beyond f16's 65504.  Unscaled hi/lo halves would be +inf (-> inf or NaN logits); the per-clip exponents keep the
    split exact, and the logits must match the float64 oracle like any other case."""
    sd = _sd(arch)
    sd["conv1.weight"] = (sd["conv1.weight"] * 8000.0).astype(np.float32)
    x = _inputs_case("logmel")
    a1 = np.maximum(model_oracle._conv3x3_relu_np(x.astype(np.float64), sd["conv1.weight"], sd["conv1.bias"]), 0)
    assert a1.max() > 2 * 65504 and model_oracle.pooled_features_np(x, sd).max() > 65504    # the case does leave the f16 range
    _check("guard:conv1x8000", arch, sd, x, dev, conv_math)
    # and inputs beyond the f16 range themselves
    _check("guard:x1e7", arch, _sd(arch), (x * 1e7 / 80).astype(np.float32), dev, conv_math)


def test_head_rows_beyond_f16_range(dev, conv_math):
    from wakeword_jupyterlab_amd import ops
    sd = _sd("simple")
    packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
    r = _rng(3)
    pooled = (np.abs(r.standard_normal((40, 64))) * (10.0 ** r.uniform(-6, 7, (40, 1)))).astype(np.float32)   # rows from 1e-6 to 1e7
    y = ops.lstm_fc(torch.from_numpy(pooled).to(dev), packed, 2).cpu().numpy()
    ref = model_oracle.head_np(pooled, sd)
    err = float(np.abs(y - ref).max())
    _WORST[f"simple/{conv_math}/head_rows_1e-6..1e7"] = err
    assert np.isfinite(y).all() and err <= LOGIT_TOL


def _f16_round(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("case", ["default", "lognormal", "outlier_x1000", "tiny_channels"])
def test_head_on_unsaturated_rows(dev, conv_math, case):
    """The LSTM gate GEMMs where the gates are NOT pinned at +-1: pooled rows scaled so that layer 0's largest |pre-activation|
    is ~1.5.  The logits then carry the GEMM's error (self-check: the same rows rounded to plain f16 -- one MFMA per block --
    move the float64 logits by >= 10x the bound asserted here)."""
    from wakeword_jupyterlab_amd import ops
    sd = _sd("simple") if case == "default" else _weights_case("simple", case)
    r = _rng(11)
    pooled = np.abs(r.standard_normal((64, 64))).astype(np.float64) + 0.05
    w0 = sd["lstm.weight_ih_l0"].astype(np.float64)
    b0 = sd["lstm.bias_ih_l0"].astype(np.float64) + sd["lstm.bias_hh_l0"].astype(np.float64)
    g = pooled @ w0.T
    pooled = (pooled * (1.5 / np.abs(g).max(axis=1, keepdims=True))).astype(np.float32)
    assert np.abs(pooled.astype(np.float64) @ w0.T + b0).max() < 4.0
    ref = model_oracle.head_np(pooled, sd)
    teeth = float(np.abs(model_oracle.head_np(_f16_round(pooled), sd) - ref).max())     # what plain-f16 inputs alone would cost
    packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
    y = ops.lstm_fc(torch.from_numpy(pooled).to(dev), packed, 2).cpu().numpy()
    err = float(np.abs(y - ref).max())
    if conv_math == "f32":
        err32 = err
    else:
        ops.set_conv_math("f32")
        try:
            err32 = float(np.abs(ops.lstm_fc(torch.from_numpy(pooled).to(dev), packed, 2).cpu().numpy() - ref).max())
        finally:
            ops.set_conv_math(conv_math)
    scale = float(np.abs(ref).max())
    bound = 2 * err32 + 1e-7
    _WORST_HEAD[f"{conv_math}/{case}"] = {"err": err, "err_f32_kernel": err32, "plain_f16_inputs_would_cost": teeth, "logit_scale": scale}
    assert np.isfinite(y).all() and err <= bound, f"{case}: head logits off by {err:.3e} (fp32 kernel {err32:.3e})"
    assert teeth >= 10 * bound, f"{case}: this case cannot see the arithmetic (plain f16 would cost {teeth:.3e}, bound {bound:.3e})"


def test_nan_and_inf_inputs_propagate_like_the_reference(dev, conv_math):
    sd = _sd("simple")
    x = _inputs_case("logmel")
    x[1, 0, 10, 10] = np.nan
    x[3, 0, 5, 5] = np.inf
    y = _run("simple", sd, x, dev)
    ref = model_oracle.forward_np(x, sd)
    ok = [0, 2, 4, 5]
    assert np.abs(y[ok] - ref[ok]).max() <= LOGIT_TOL                  # clips are independent: the others are untouched
    assert np.isnan(y[1]).all() and not np.isfinite(y[3]).any()        # NaN stays NaN; inf -> inf - inf = NaN in the reference too


def test_per_thread_math_overrides_do_not_leak_between_threads(dev):
    """Two host threads with different per-thread conv arithmetics (and one with a log-mel override) launch concurrently on their own
    streams for a while: each must get, bit for bit, what a single-threaded run under that arithmetic gives, and the process default is
    untouched (the process-wide switches were unsafe for a streamer beside a batch job)."""
    import threading
    from wakeword_jupyterlab_amd import ops
    sd = _sd("simple")
    packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
    x = torch.from_numpy(_inputs_case("logmel", batch=96)).to(dev)
    pcm = torch.from_numpy(pkg.synth.make_clips(500, 24)).to(dev)
    want = {}
    for m in ("f16x3", "f16x3d", "f32"):
        ops.set_conv_math(m)
        want[m] = ops.cnn_pool(x, packed, 2).clone()
    ops.set_conv_math("f16x3")
    ops.set_logmel_math("f64")
    want_mel64 = ops.logmel(pcm, True).clone()
    ops.set_logmel_math("auto")
    want_mel = ops.logmel(pcm, True).clone()
    assert not torch.equal(want["f16x3"], want["f32"])
    torch.cuda.synchronize()
    errors = []

    def worker(mode, mel_mode):
        try:
            torch.cuda.set_device(dev)
            ops.set_conv_math_thread(mode)
            ops.set_logmel_math_thread(mel_mode)
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(60):
                    got = ops.cnn_pool(x, packed, 2)
                    mel = ops.logmel(pcm, True)
                    torch.cuda.current_stream().synchronize()
                    if not torch.equal(got, want[mode]) or not torch.equal(mel, want_mel64 if mel_mode == "f64" else want_mel):
                        errors.append(mode)
                        return
            ops.set_conv_math_thread(None)
            ops.set_logmel_math_thread(None)
        except Exception as e:                       # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=a) for a in (("f32", "f64"), ("f16x3d", None), ("f16x3", None))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert ops.get_conv_math() == "f16x3" and ops.get_logmel_math() == "auto"          # this thread and the process default: untouched
    assert torch.equal(ops.cnn_pool(x, packed, 2), want["f16x3"])


# ---------------------------------------------------------------------------------------------------------------
# single clips of any length (the stride argument of the C ABI is irrelevant for one row)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 3, 511, 8001, 15999, 16000])
def test_single_clip_of_any_length(dev, n):
    from wakeword_jupyterlab_amd import ops
    from wakeword_jupyterlab_amd.audio import AudioProcessor
    x = pkg.synth.make_clips(300, 1)[:, :n].copy()
    ref = mel_oracle.logmel_batch(x, normalize=True)
    out = ops.logmel(torch.from_numpy(x).to(dev), True).cpu().numpy()
    assert np.abs(out - ref).max() <= 1e-4
    mel = AudioProcessor().audio_to_mel(x[0] / np.abs(x[0]).max())     # the reference call order: normalize_audio first
    assert mel.shape == (80, 32) and np.abs(mel - ref[0, 0]).max() <= 1e-4
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = pkg.SimpleWakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    with torch.no_grad():
        y = m.forward_pcm(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.abs(y - model_oracle.forward_np(ref, sd)).max() <= LOGIT_TOL


# ---------------------------------------------------------------------------------------------------------------
# an expired in-kernel wait poisons the launch
# ---------------------------------------------------------------------------------------------------------------
_POISON_CHILD = r"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.environ["WW_ROOT"])
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import ops, _native as nat
assert nat.LIB_PATH.endswith("libwakeword_amd_spin1.so"), nat.LIB_PATH
dev = torch.device("cuda", 0)
res = {}
x = torch.from_numpy((np.random.default_rng(0).standard_normal((700, 1, 80, 32)) * 15 - 35).astype(np.float32)).to(dev)
for math in ("f16x3", "f16x3d"):                     # cnn2w / cnn3w (Winograd) and cnn2h16 / cnn3h (direct)
    ops.set_conv_math(math)
    for arch, n_conv in (("simple", 2), ("full", 3)):
        packed = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict(arch, seed=3))).to(dev)
        pooled = ops.cnn_pool(x, packed, n_conv)
        torch.cuda.synchronize()
        res[math + "/" + arch] = {"nan_rows": int(torch.isnan(pooled).all(dim=1).sum()), "finite_rows": int(torch.isfinite(pooled).all(dim=1).sum()),
                                  "rows": int(pooled.shape[0])}
# the training forwards (cnn2w_kernel<2>, cnn2w_kernel<3> + cnn3w_kernel<true>: the instantiations with the bit-image outputs)
ops.set_conv_math("f16x3"); ops.set_train_math("f16x3")
for arch in ("simple", "full"):
    m = (pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in pkg.synth.make_state_dict(arch, seed=3).items()})
    m = m.to(dev).train()
    with torch.no_grad():
        logits = m(x)
    torch.cuda.synchronize()
    res["train/" + arch] = {"nan_rows": int(torch.isnan(logits).all(dim=1).sum()), "finite_rows": int(torch.isfinite(logits).all(dim=1).sum()),
                            "rows": int(logits.shape[0])}
res["timeouts"] = int(nat.lib.ww_sync_timeouts())
print("RESULT " + json.dumps(res))
"""


def test_expired_wait_poisons_the_outputs_with_nan():
    """Diagnostic twin of the library (csrc/Makefile: -DWW_FLAG_SPINS=1): every bounded wait of the conv kernel gives up after
    one poll, i.e. the producer / consumer protocol is broken on purpose.  Every workgroup that saw an expired wait must
    overwrite ALL its pooled features with NaN and the counter must say so; rows are either all-NaN or all-finite.  Covered: the
    Winograd and the direct split-precision kernels of both models, and the training forwards (the instantiations that also write
    the ReLU bit images), whose logits must then be NaN for the poisoned clips."""
    twin = os.path.join(ROOT, "wakeword-jupyterlab_amd", "csrc", "build", "libwakeword_amd_spin1.so")
    assert os.path.exists(twin), "build the diagnostic twin: make -C wakeword-jupyterlab_amd/csrc"
    env = dict(os.environ, WW_LIB_OVERRIDE=twin, WW_ROOT=ROOT, WW_CONV_MATH="f16x3")
    p = subprocess.run([sys.executable, "-c", _POISON_CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert res["timeouts"] > 0
    for key in ("f16x3/simple", "f16x3/full", "f16x3d/simple", "f16x3d/full", "train/simple", "train/full"):
        r = res[key]
        assert r["nan_rows"] > 0, (key, r)                                   # the broken run cannot be consumed silently
        assert r["nan_rows"] + r["finite_rows"] == r["rows"], (key, r)        # never a half-written clip


_X32_CHILD = r"""
import json, os, sys
sys.path.insert(0, os.environ["WW_ROOT"])
import numpy as np, torch
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat, ops
from oracle import model_oracle
dev = torch.device("cuda", 0)
sd = pkg.synth.make_state_dict("simple", seed=1234)
packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
pcm = torch.from_numpy(pkg.synth.make_clips(0, 24)).to(dev)
mel = ops.logmel(pcm, True)
res = {}
for width in (32, 31, 5):
    m = mel[..., :width].contiguous()
    a = ops.cnn_pool(m, packed, 2); b = ops.cnn_pool(m, packed, 2)
    torch.cuda.synchronize()
    ref = model_oracle.pooled_features_np(m.cpu().numpy().astype(np.float64), sd)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    res[str(width)] = {"err": float((np.abs(a.cpu().numpy() - ref) / scale).max()), "repeat": bool((a == b).all())}
    if width == 32:
        a_full = a
big = mel.repeat(46, 1, 1, 1)[:1100].contiguous()     # > 4 clips per workgroup: rings and exchange slots wrap; every 24th row is the same image
pb = ops.cnn_pool(big, packed, 2)
torch.cuda.synchronize()
res["tiled_equal"] = bool((pb[:24] == pb[24:48]).all()) and bool((pb[:24] == pb[1056:1080]).all()) and bool((pb[:24] == a_full).all())
res["timeouts"] = int(nat.lib.ww_sync_timeouts())
print("RESULT " + json.dumps(res))
"""


def test_conv2_on_32x32x16_tiles_is_a_parity_correct_alternative():
    """cnn2x_kernel (WW_K2_FORM=x32, read once per process: hence the child): conv2 of the 2-conv model with the consumers on
    v_mfma_f32_32x32x16_f16 and the four Winograd xi split over wave pairs that exchange an accumulator through LDS.  Round 4 measured it 20 %
    slower than the shipped 16x16x32 form (profiles/r04_k2_census.txt) and it is not the default; it stays in the library as the
    measured alternative, so it must keep meeting the same bound as the shipped kernel: pooled features within 1e-6 of the float64
    oracle relative to the clip's largest feature, bitwise repeatable, no expired wait, for full and ragged widths."""
    env = dict(os.environ, WW_ROOT=ROOT, WW_K2_FORM="x32")
    p = subprocess.run([sys.executable, "-c", _X32_CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert res["timeouts"] == 0
    assert res["tiled_equal"]
    for width in ("32", "31", "5"):
        assert res[width]["err"] <= 1e-6, (width, res[width])
        assert res[width]["repeat"], width
