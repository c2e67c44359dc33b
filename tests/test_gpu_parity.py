"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference-generated goldens.

Tolerances are the ones BASELINE.json's north_star states: 1e-4 dB on log-mel values, 1e-3 on logits.
"""
import os

import numpy as np
import pytest
import torch

import wakeword_jupyterlab_amd as pkg
from oracle import mel_oracle, model_oracle

pytestmark = pytest.mark.gpu

MEL_TOL = 1e-4      # dB, north_star
LOGIT_TOL = 1e-3    # north_star


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need the MI355X"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def ops():
    from wakeword_jupyterlab_amd import ops as o
    return o


@pytest.fixture(params=["f16x3", "f16x3d", "f32"], autouse=True)
def conv_math(request, ops):
    """Every test runs under all three conv arithmetics: the default split-precision f16x3 kernels (conv2 of the 2-conv model as 1-D
    Winograd), f16x3 with every conv direct, and the exact-f32 MFMA kernels."""
    ops.set_conv_math(request.param)
    yield request.param
    ops.set_conv_math("f16x3")


@pytest.fixture(scope="module")
def clips64():
    return pkg.synth.make_clips(0, 64)          # BASELINE config 1: 64 synthetic 1 s clips


@pytest.fixture(scope="module")
def ref_mel64(clips64):
    return mel_oracle.logmel_batch(clips64, normalize=True)


def _model(arch, sd, dev):
    m = pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev).eval()


# ------------------------------------------------------------------------------------------------ K1
def test_logmel_matches_oracle_config1(ops, dev, clips64, ref_mel64):
    out = ops.logmel(torch.from_numpy(clips64).to(dev), True).cpu().numpy()
    assert out.shape == (64, 1, 80, 32) and out.dtype == np.float32
    err = np.abs(out - ref_mel64).max(axis=(1, 2, 3))
    assert err.max() <= MEL_TOL, f"max |log-mel err| = {err.max():.3e} dB at clip {err.argmax()}"
    # power_to_db(ref=np.max, top_db=80): per-clip max exactly 0, floor -80
    assert np.all(out.max(axis=(1, 2, 3)) == 0.0) and out.min() >= -80.0


def test_logmel_without_normalisation_and_float64_budget(ops, dev, clips64):
    x = clips64[:8]
    out = ops.logmel(torch.from_numpy(x).to(dev), False).cpu().numpy()
    ref = mel_oracle.logmel_batch(x, normalize=False)
    assert np.abs(out - ref).max() <= MEL_TOL
    for i in range(8):                     # and against the all-float64 evaluation
        assert np.abs(out[i, 0] - mel_oracle.logmel_f64(x[i], normalize=False)).max() <= MEL_TOL


@pytest.mark.parametrize("n", [1, 3, 511, 9000, 15999, 16000])
def test_logmel_short_clips_are_right_padded(ops, dev, n):
    x = pkg.synth.make_clips(100, 3)[:, :n].copy()
    out = ops.logmel(torch.from_numpy(x).to(dev), True).cpu().numpy()
    ref = mel_oracle.logmel_batch(x, normalize=True)
    assert np.abs(out - ref).max() <= MEL_TOL


def test_logmel_silent_clip_is_nan_like_reference_and_does_not_leak(ops, dev, clips64):
    x = clips64[:4].copy()
    x[2] = 0.0
    out = ops.logmel(torch.from_numpy(x).to(dev), True).cpu().numpy()
    assert np.isnan(out[2]).all()                                  # x / max|x| = 0/0 (wakeword_training_script.py:73-76)
    ref = mel_oracle.logmel_batch(x[[0, 1, 3]], normalize=True)
    assert np.abs(out[[0, 1, 3]] - ref).max() <= MEL_TOL
    flat = ops.logmel(torch.from_numpy(x[2:3]).to(dev), False).cpu().numpy()
    assert np.all(flat == 0.0)                                     # amin clamp: every bin at the reference level


def _clean_signals():
    t = np.arange(16000) / 16000.0
    return np.stack([np.sin(2 * np.pi * 440.0 * t), np.sin(2 * np.pi * 3999.5 * t) * 0.01,
                     (np.arange(16000) == 8000).astype(np.float64), np.sign(np.sin(2 * np.pi * 100 * t)),
                     np.sin(2 * np.pi * 7400.0 * t) + 1e-3 * np.sin(2 * np.pi * 250.0 * t),
                     np.sin(2 * np.pi * 150.0 * t) * (np.arange(16000) < 6000)]).astype(np.float32)


def _real_world_clean_signals():
    """Families real WAV files produce and the six above do not: 16-bit-quantised content (a quantisation floor
    ~98 dB under full scale), a DC offset with weak content, energy at Nyquist (bin 1024 has zero mel weight: the frame energy seen
    through the mel bands under-counts it), hard clipping, chirps -- and a quantised tone that is also very quiet (few LSBs)."""
    t = np.arange(16000) / 16000.0
    q16 = lambda v: np.round(np.clip(v, -1, 1 - 2.0 ** -15) * 32768.0) / 32768.0       # what a PCM-16 file holds  # noqa: E731
    chirp = lambda f0, f1: np.sin(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t))      # noqa: E731
    sigs = [
        q16(0.5 * np.sin(2 * np.pi * 440.0 * t)),                                       # quantised tone
        q16(0.3 * np.sin(2 * np.pi * 1234.5 * t) + 0.2 * np.sin(2 * np.pi * 310.0 * t)),  # quantised tone pair
        q16(3.0 / 32768.0 * np.sin(2 * np.pi * 700.0 * t)),                             # three LSBs of signal
        0.5 + 1e-3 * np.sin(2 * np.pi * 600.0 * t),                                     # DC offset, weak content
        q16(0.25 + 2e-3 * np.sin(2 * np.pi * 2500.0 * t)),                              # the same, quantised
        0.8 * np.cos(np.pi * np.arange(16000)) + 1e-3 * np.sin(2 * np.pi * 900.0 * t),  # Nyquist tone + weak content
        0.8 * np.cos(np.pi * np.arange(16000)),                                         # Nyquist tone alone (mel sees only leakage)
        np.clip(3.0 * np.sin(2 * np.pi * 220.0 * t), -1.0, 1.0),                        # hard clipping
        q16(np.clip(1.7 * np.sin(2 * np.pi * 523.25 * t) * (t > 0.2), -1.0, 1.0)),      # clipped, gated, quantised
        chirp(100.0, 7000.0),                                                           # fast chirp over the band
        chirp(3000.0, 3050.0) * 0.05,                                                   # slow, quiet chirp
        q16(0.4 * chirp(200.0, 3800.0)),                                                # quantised chirp
    ]
    return np.stack(sigs).astype(np.float32)


@pytest.mark.parametrize("mode", ["auto", "f64", "f32"])
def test_logmel_real_world_clean_families(ops, dev, mode):
    """auto and f64 must meet north_star's 1e-4 dB on every family; the plain-f32 run is recorded for the calibration (it may miss it:
    that is what auto mode is for) and must stay finite inside [-80, 0]."""
    x = _real_world_clean_signals()
    ref = mel_oracle.logmel_batch(x, normalize=True)
    ops.set_logmel_math(mode)
    try:
        out = ops.logmel(torch.from_numpy(x).to(dev), True).cpu().numpy()
    finally:
        ops.set_logmel_math("auto")
    err = np.abs(out - ref).max(axis=(1, 2, 3))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        import json
        with open(os.path.join(out_dir, f"clean_families_{mode}.json"), "w") as f:
            json.dump({"mode": mode, "max_abs_err_dB_per_signal": [float(e) for e in err]}, f)
    assert np.isfinite(out).all() and out.min() >= -80.0 and np.all(out.max(axis=(1, 2, 3)) == 0.0)
    if mode != "f32":
        assert err.max() <= MEL_TOL, err


@pytest.mark.parametrize("mode", ["auto", "f64"])
def test_logmel_noise_free_signals_meet_the_tolerance_over_the_whole_range(ops, dev, mode):
    """Pure tones, an impulse, a square wave, a tone pair 60 dB apart, a gated tone: mel bands 60-80 dB below the clip's peak sit
    on a float32 FFT's rounding floor (the reference's FFT is float64: numpy.fft.rfft).  The default (auto) mode redoes such
    clips in float64; north_star's 1e-4 dB holds over the whole [-80, 0] dB range, no exception."""
    x = _clean_signals()
    ref = mel_oracle.logmel_batch(x, normalize=True)
    ops.set_logmel_math(mode)
    try:
        out = ops.logmel(torch.from_numpy(x).to(dev), True).cpu().numpy()
        single = np.concatenate([ops.logmel(torch.from_numpy(x[i:i + 1]).to(dev), True).cpu().numpy() for i in range(len(x))])
    finally:
        ops.set_logmel_math("auto")
    assert np.abs(out - ref).max() <= MEL_TOL, np.abs(out - ref).max(axis=(1, 2, 3))
    assert np.array_equal(out, single)                              # 8-wave (small batch) and 4-wave forms agree
    assert np.all(out.max(axis=(1, 2, 3)) == 0.0) and out.min() >= -80.0


@pytest.mark.parametrize("n", [1, 511, 9000, 16000])
def test_logmel_f64_mode_short_clips_and_strides(ops, dev, n):
    x = pkg.synth.make_clips(100, 3)[:, :n].copy()
    x[1] *= 0.01
    ops.set_logmel_math("f64")
    try:
        out = ops.logmel(torch.from_numpy(x).to(dev), True).cpu().numpy()
        raw = ops.logmel(torch.from_numpy(x).to(dev), False).cpu().numpy()
        one = ops.logmel(torch.from_numpy(x[2:3]).to(dev), True).cpu().numpy()
    finally:
        ops.set_logmel_math("auto")
    assert np.abs(out - mel_oracle.logmel_batch(x, normalize=True)).max() <= MEL_TOL
    assert np.abs(raw - mel_oracle.logmel_batch(x, normalize=False)).max() <= MEL_TOL
    assert np.array_equal(one[0], out[2])


def test_logmel_f32_mode_floor_and_auto_leaves_noisy_clips_alone(ops, dev, clips64, ref_mel64):
    """The float32 kernel alone: within tolerance down to -60 dB on noise-free signals, ~3e-4 dB on its rounding floor below
    (recorded, not the shipped default).  Auto mode must not touch clips with a broadband floor: the benchmark's clips
    come out bit-identical to f32 mode."""
    x = _clean_signals()
    ref = mel_oracle.logmel_batch(x, normalize=True)
    ops.set_logmel_math("f32")
    try:
        out32 = ops.logmel(torch.from_numpy(x).to(dev), True).cpu().numpy()
        noisy32 = ops.logmel(torch.from_numpy(clips64).to(dev), True).cpu().numpy()
    finally:
        ops.set_logmel_math("auto")
    err = np.abs(out32 - ref)
    assert err[ref >= -60.0].max() <= MEL_TOL and err.max() <= 2e-3
    noisy_auto = ops.logmel(torch.from_numpy(clips64).to(dev), True).cpu().numpy()
    assert np.array_equal(noisy_auto, noisy32) and np.abs(noisy_auto - ref_mel64).max() <= MEL_TOL
    assert ops.get_logmel_math() == "auto"
    with pytest.raises(ValueError):
        ops.set_logmel_math("f16")


def test_logmel_strided_and_unaligned_inputs(ops, dev, clips64, ref_mel64):
    big = torch.zeros(8, 16004, device=dev)
    big[:, 2:16002] = torch.from_numpy(clips64[:8]).to(dev)
    out = ops.logmel(big[:, 2:16002], True).cpu().numpy()          # misaligned view -> the op re-lays it out
    assert np.abs(out - ref_mel64[:8]).max() <= MEL_TOL
    out = ops.logmel(torch.from_numpy(clips64[:8]).to(dev)[::2], True).cpu().numpy()
    assert np.abs(out - ref_mel64[:8:2]).max() <= MEL_TOL


def test_logmel_empty_batch_and_bad_arguments(ops, dev):
    assert ops.logmel(torch.zeros(0, 16000, device=dev), True).shape == (0, 1, 80, 32)
    with pytest.raises(ValueError):
        ops.logmel(torch.zeros(2, 16001, device=dev), True)
    with pytest.raises(RuntimeError):
        ops.logmel(torch.zeros(2, 16000), True)                    # CPU tensor: no CPU path
    with pytest.raises(TypeError):
        ops.logmel(torch.zeros(2, 16000, device=dev, dtype=torch.float64), True)


# ------------------------------------------------------------------------------------------------ K2 + K3
@pytest.mark.parametrize("tag", ["32", "31"])
def test_simple_model_matches_reference_goldens(ops, dev, golden_simple, tag):
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
    x = torch.from_numpy(golden_simple["x" + tag]).to(dev)
    pooled = ops.cnn_pool(x, packed, 2).cpu().numpy()
    logits = ops.cnn_lstm_forward(x, packed, 2).cpu().numpy()
    assert np.abs(pooled - golden_simple["pooled" + tag]).max() <= 1e-4
    assert np.abs(logits - golden_simple["logits" + tag]).max() <= LOGIT_TOL
    head = ops.lstm_fc(torch.from_numpy(golden_simple["pooled" + tag]).to(dev), packed, 2).cpu().numpy()
    assert np.abs(head - golden_simple["logits" + tag]).max() <= 1e-5


@pytest.mark.parametrize("tag", ["32", "31"])
def test_full_model_matches_reference_class_goldens(ops, dev, golden_full, tag):
    """3-conv WakewordModel against outputs of the reference's own class (wakeword_training_script.py:141-184)."""
    sd = pkg.synth.make_state_dict("full", seed=1234)
    packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
    x = torch.from_numpy(golden_full["x" + tag]).to(dev)
    pooled = ops.cnn_pool(x, packed, 3).cpu().numpy()
    logits = ops.cnn_lstm_forward(x, packed, 3).cpu().numpy()
    assert np.abs(pooled - golden_full["pooled" + tag]).max() <= 1e-4
    assert np.abs(logits - golden_full["logits" + tag]).max() <= LOGIT_TOL
    head = ops.lstm_fc(torch.from_numpy(golden_full["pooled" + tag]).to(dev), packed, 3).cpu().numpy()
    assert np.abs(head - golden_full["logits" + tag]).max() <= 1e-5


_ORACLE_CACHE = {}


def _oracle_logits(arch, width, batch, x, sd):
    """Closed-form numpy oracle for small batches; its torch restatement (pinned to it in tests/test_oracle_model.py) for the
    large ones, computed once for both conv maths."""
    key = (arch, width, batch)
    if key not in _ORACLE_CACHE:
        if batch <= 64:
            _ORACLE_CACHE[key] = model_oracle.forward_np(x, sd)
        else:
            with torch.no_grad():
                _ORACLE_CACHE[key] = model_oracle.torch_module_from_state_dict(sd)(torch.from_numpy(x)).numpy()
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("arch,width,batch", [("simple", 32, 33), ("simple", 17, 5), ("simple", 1, 2), ("simple", 31, 300),
                                              ("simple", 8, 258), ("full", 32, 9), ("full", 31, 3), ("full", 16, 4),
                                              ("full", 1, 2), ("full", 23, 258)])
def test_model_module_matches_oracle(dev, arch, width, batch):
    sd = pkg.synth.make_state_dict(arch, seed=7)
    x = (pkg.synth.normal(11, batch * 80 * width).astype(np.float32).reshape(batch, 1, 80, width) * 15 - 35)
    m = _model(arch, sd, dev)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert y.shape == (batch, 2)
    assert np.abs(y - _oracle_logits(arch, width, batch, x, sd)).max() <= LOGIT_TOL


def test_state_dict_round_trip_and_repack_on_update(dev, golden_simple):
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = _model("simple", sd, dev)
    x = torch.from_numpy(golden_simple["x32"]).to(dev)
    with torch.no_grad():
        y0 = m(x).cpu().numpy()
    assert np.abs(y0 - golden_simple["logits32"]).max() <= LOGIT_TOL
    assert set(m.state_dict().keys()) == set(sd.keys())
    # weight_hh / forget-gate rows are dead; a live weight must change the output (cache invalidation)
    with torch.no_grad():
        m.lstm.weight_hh_l0.fill_(3.0)
        assert np.array_equal(m(x).cpu().numpy(), y0)
        m.fc.bias.add_(1.0)
        assert np.abs(m(x).cpu().numpy() - (y0 + 1.0)).max() <= 1e-5
    sd2 = pkg.synth.make_state_dict("simple", seed=99)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()})
    with torch.no_grad():
        y2 = m(x).cpu().numpy()
    assert np.abs(y2 - model_oracle.forward_np(golden_simple["x32"], sd2)).max() <= LOGIT_TOL


def test_cpu_inputs_and_unsupported_shapes_fail_loudly(dev):
    m = pkg.SimpleWakewordModel().to(dev)
    m.eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 80, 32))
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 1, 80, 33, device=dev))
    with pytest.raises(RuntimeError):
        pkg.SimpleWakewordModel().eval()(torch.zeros(1, 1, 80, 32, device=dev))   # parameters on the CPU


# ------------------------------------------------------------------------------------------------ end to end
def test_forward_pcm_config1_end_to_end(dev, clips64, ref_mel64):
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = _model("simple", sd, dev)
    with torch.no_grad():
        y = m.forward_pcm(torch.from_numpy(clips64).to(dev)).cpu().numpy()
    ref = model_oracle.forward_np(ref_mel64, sd)
    assert np.abs(y - ref).max() <= LOGIT_TOL
    tm = model_oracle.torch_module_from_state_dict(sd)            # and against torch's own CPU layers
    with torch.no_grad():
        yt = tm(torch.from_numpy(ref_mel64)).numpy()
    assert np.abs(y - yt).max() <= LOGIT_TOL


def test_full_batch_properties_4096(ops, dev):
    """BASELINE's full size: properties that need no oracle run (it would take minutes on the CPU)."""
    base = pkg.synth.make_clips(0, 64)
    reps = 4096 // 64
    gains = (1.0 - 0.5 * (np.arange(reps) / reps)).astype(np.float32)
    pcm = torch.from_numpy((base[None] * gains[:, None, None]).reshape(4096, 16000)).to(dev)
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = _model("simple", sd, dev)
    with torch.no_grad():
        mel = ops.logmel(pcm, True)
        y = m.forward_pcm(pcm)
        y_again = m.forward_pcm(pcm)
        perm = torch.randperm(4096, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
        y_perm = m.forward_pcm(pcm[perm])
    assert torch.all(mel.amax(dim=(1, 2, 3)) == 0.0) and float(mel.min()) >= -80.0
    assert torch.equal(y, y_again)                                 # deterministic
    assert torch.equal(y[perm], y_perm)                            # clips are independent: batch position is irrelevant
    # gain invariance of the whole path (per-clip peak normalisation): every replica of a clip agrees
    yr = y.reshape(reps, 64, 2)
    assert float((yr - yr[0:1]).abs().max()) <= LOGIT_TOL
    ref = model_oracle.forward_np(mel_oracle.logmel_batch(base[:8]), sd)
    assert np.abs(yr[0, :8].cpu().numpy() - ref).max() <= LOGIT_TOL


@pytest.mark.parametrize("arch", ["simple", "full"])
@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 511, 517, 1030, 2049])
def test_ragged_batch_sizes_agree_with_small_batches(ops, dev, arch, n):
    # persistent kernels loop over clips with grid = resident workgroups: batch sizes around the CU count (256) and
    # not divisible by anything must give, clip for clip, exactly what small batches give
    base = pkg.synth.make_clips(200, 32)
    pcm = torch.from_numpy(np.tile(base, (n // 32 + 1, 1))[:n] * np.linspace(1.0, 0.5, n, dtype=np.float32)[:, None]).to(dev)
    m = _model(arch, pkg.synth.make_state_dict(arch, seed=5), dev)
    with torch.no_grad():
        whole = m.forward_pcm(pcm)
        parts = torch.cat([m.forward_pcm(pcm[s:s + 37]) for s in range(0, n, 37)])
        mel_whole = ops.logmel(pcm, True)
        mel_parts = torch.cat([ops.logmel(pcm[s:s + 37], True) for s in range(0, n, 37)])
    assert torch.equal(mel_whole, mel_parts)
    assert torch.equal(whole, parts)


def test_custom_ops_are_registered(ops, dev):
    for name in ("logmel", "cnn_pool", "lstm_fc", "cnn_lstm_forward", "forward_pcm"):
        assert hasattr(torch.ops.wakeword_amd, name)
    out = torch.ops.wakeword_amd.logmel(torch.from_numpy(pkg.synth.make_clips(5, 2)).to(dev), True)
    assert out.shape == (2, 1, 80, 32)


def test_compiled_ops_agree_with_the_raw_launches_and_trace_under_torch_compile(ops, dev):
    """The compiled operators (csrc/ww_torch_ops.cpp) against the ctypes launches they replaced, bit for bit, including the inputs that
    need a copy (strided / unaligned PCM, ragged widths); then a function that calls them traced by torch.compile (backend aot_eager: the
    graph is captured with FakeTensors -- the Meta kernels -- and run through the real kernels; no code generation involved)."""
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    packed = torch.from_numpy(ops.pack_state_dict(sd)).to(dev)
    pcm = torch.from_numpy(pkg.synth.make_clips(20, 9)).to(dev)
    wide = torch.zeros(9, 16004, device=dev); wide[:, 3:16003] = pcm[:, :16000]
    for x in (pcm, pcm[:, :15999], wide[:, 3:16003], pcm[::2], pcm[1:2, :777]):
        assert torch.equal(torch.ops.wakeword_amd.logmel(x, True), ops._logmel_impl(x, True))
        assert torch.equal(torch.ops.wakeword_amd.forward_pcm(x, packed, 2, True), ops._forward_pcm_impl(x, packed, 2, True))
    mel = ops.logmel(pcm, True)
    for w in (32, 31, 5):
        m = mel[..., :w]
        assert torch.equal(torch.ops.wakeword_amd.cnn_pool(m, packed, 2), ops._cnn_pool_impl(m, packed, 2))
        assert torch.equal(torch.ops.wakeword_amd.cnn_lstm_forward(m, packed, 2), ops._cnn_lstm_forward_impl(m, packed, 2))
    pooled = ops.cnn_pool(mel, packed, 2)
    assert torch.equal(torch.ops.wakeword_amd.lstm_fc(pooled, packed, 2), ops._lstm_fc_impl(pooled, packed, 2))

    def path(p):
        return torch.softmax(torch.ops.wakeword_amd.forward_pcm(p, packed, 2, True), dim=1)[:, 1]
    traced = torch.compile(path, backend="aot_eager", fullgraph=True)
    assert torch.equal(traced(pcm), path(pcm))


@pytest.mark.parametrize("arch,n,launches", [("full", 777, 1500), ("simple", 1000, 3000)])
def test_conv_stack_is_bitwise_repeatable_back_to_back(ops, dev, arch, n, launches):
    """Soak: the producer / consumer hand-off inside the conv kernel is counter-synchronised (no workgroup barrier); a hole in
    that protocol shows up as a rare run-to-run difference (one was found this way: a single running total for consumers that
    are not in lock step).  scripts/soak.py runs the long version."""
    n_conv = 2 if arch == "simple" else 3
    packed = torch.from_numpy(ops.pack_state_dict(pkg.synth.make_state_dict(arch, seed=3))).to(dev)
    pcm = torch.from_numpy(pkg.synth.make_clips_tiled(0, n, unique=64)).to(dev)
    mel = ops.logmel(pcm, True)
    ref = ops.cnn_pool(mel, packed, n_conv).clone()
    bad = 0
    for _ in range(launches // 50):
        outs = [ops.cnn_pool(mel, packed, n_conv) for _ in range(50)]
        bad += sum(not torch.equal(o, ref) for o in outs)
    assert bad == 0


def _random_noise_free_signals(n_sig, seed):
    """Seeded noise-free signals of the kinds digitally clean corpora hold: multi-tones, AM-FM carriers, chirps, decaying resonances of
    impulse trains, square / clipped waves -- each optionally gated (digital silence), with a DC offset, a random level, quantised to a
    random bit depth (8..24, or left in float) and of random length (right zero-padded by the path itself)."""
    r = np.random.default_rng(seed)
    t = np.arange(16000) / 16000.0
    sigs, kinds = [], []
    for _ in range(n_sig):
        kind = int(r.integers(0, 6))
        if kind == 0:                                                   # 1..6 partials, levels over 80 dB
            x = np.zeros(16000)
            for _p in range(int(r.integers(1, 7))):
                f = float(np.exp(r.uniform(np.log(30.0), np.log(7990.0))))
                x += 10.0 ** r.uniform(-4.0, 0.0) * np.sin(2 * np.pi * f * t + r.uniform(0, 2 * np.pi))
        elif kind == 1:                                                 # AM-FM carrier
            fc, fm, dev_hz = r.uniform(100.0, 7000.0), r.uniform(0.5, 40.0), r.uniform(0.0, 300.0)
            am_f, am_d = r.uniform(0.5, 30.0), r.uniform(0.0, 1.0)
            x = (1.0 + am_d * np.sin(2 * np.pi * am_f * t)) * np.sin(2 * np.pi * fc * t + dev_hz / fm * np.sin(2 * np.pi * fm * t))
        elif kind == 2:                                                 # chirp
            f0, f1 = r.uniform(30.0, 7900.0), r.uniform(30.0, 7900.0)
            x = np.sin(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t))
        elif kind == 3:                                                 # impulse train through a two-pole resonator
            x = np.zeros(16000)
            period = int(r.integers(40, 4000))
            x[int(r.integers(0, period))::period] = 1.0
            w0, rad = 2 * np.pi * r.uniform(80.0, 7000.0) / 16000.0, r.uniform(0.9, 0.9995)
            from scipy.signal import lfilter
            x = lfilter([1.0], [1.0, -2.0 * rad * np.cos(w0), rad * rad], x)
        elif kind == 4:                                                 # square / hard-clipped tone
            f = r.uniform(40.0, 3000.0)
            x = np.clip(r.uniform(1.0, 8.0) * np.sin(2 * np.pi * f * t), -1.0, 1.0)
        else:                                                           # tone pair far apart in level and frequency
            x = np.sin(2 * np.pi * r.uniform(3000.0, 7900.0) * t) + 10.0 ** r.uniform(-4.5, -1.0) * np.sin(2 * np.pi * r.uniform(60.0, 900.0) * t)
        if r.random() < 0.35:                                           # gate: digital silence before / after
            a, b = sorted(int(v) for v in r.integers(0, 16000, 2))
            if b - a > 600:
                x = x * ((np.arange(16000) >= a) & (np.arange(16000) < b))
        x = x / max(1e-30, np.abs(x).max()) * 10.0 ** r.uniform(-2.5, 0.0)
        if r.random() < 0.3:
            x = x * 0.5 + r.uniform(-0.5, 0.5)                          # DC offset
        if r.random() < 0.7:                                            # what a PCM file of that depth holds
            bits = int(r.integers(8, 25))
            x = np.round(np.clip(x, -1.0, 1.0 - 2.0 ** (1 - bits)) * 2.0 ** (bits - 1)) / 2.0 ** (bits - 1)
        n = 16000 if r.random() < 0.7 else int(r.integers(400, 16000))
        x = x[:n].astype(np.float32)
        if not np.any(x):
            x[0] = np.float32(2.0 ** -12)
        sigs.append(np.pad(x, (0, 16000 - n)) if n < 16000 else x)      # the zero pad is part of the clip either way
        kinds.append(kind)
    return np.stack(sigs).astype(np.float32), np.array(kinds)


def test_logmel_auto_mode_on_2048_random_noise_free_signals(ops, dev):
    """auto mode's rounding-floor test rests on a threshold calibrated on ~150 hand-picked signals, and round 3 found
    a hole by accident (a pure Nyquist tone).  One seeded sweep over 2,048 random noise-free signals: auto mode must meet north_star's
    1e-4 dB against the oracle on every one.  Recorded (gpurun_out/auto_random.json -> profiles/): per family the worst error, how many
    clips were redone in float64, and -- for the clips auto mode LEFT in float32 -- the smallest live-band ratio P_b / (wmax_b E_frame),
    i.e. how close an unmarked clip came to the threshold kFloorRatio = 1e-5."""
    import json
    x, kinds = _random_noise_free_signals(2048, seed=20251005)
    ref = mel_oracle.logmel_batch(x, normalize=True)
    xd = torch.from_numpy(x).to(dev)
    auto = ops.logmel(xd, True).cpu().numpy()
    ops.set_logmel_math("f32")
    try:
        f32 = ops.logmel(xd, True).cpu().numpy()
    finally:
        ops.set_logmel_math("auto")
    err = np.abs(auto - ref).max(axis=(1, 2, 3))
    err32 = np.abs(f32 - ref).max(axis=(1, 2, 3))
    redone = ~np.all(auto == f32, axis=(1, 2, 3))
    # the margin of the clips left alone: smallest P_b / (wmax_b E) over live bands, from the float64 evaluation of the oracle's pipeline
    basis = mel_oracle.mel_filterbank().astype(np.float64)
    wmax = basis.max(axis=1)
    win = mel_oracle.hann_window()[:, None]
    worst_ratio = np.inf
    for i in np.nonzero(~redone)[0]:
        y = (x[i] / np.float32(np.abs(x[i]).max())).astype(np.float64)          # normalize_audio in float32, like the path
        S = np.abs(np.fft.rfft(win * mel_oracle.frame_signal(y), axis=0)) ** 2           # [1025, 32]
        P = basis @ S
        E = S.sum(axis=0)
        live = (P > P.max() * 1e-8) & (P > 1e-10)
        ratio = np.where(live, P / wmax[:, None] / np.maximum(E[None, :], 1e-300), np.inf)
        worst_ratio = min(worst_ratio, float(ratio.min()))
    rec = {"signals": int(len(x)), "seed": 20251005, "tolerance_dB": MEL_TOL, "auto_max_err_dB": float(err.max()),
           "f32_max_err_dB": float(err32.max()), "f32_over_tolerance": int((err32 > MEL_TOL).sum()), "redone_in_float64": int(redone.sum()),
           "f32_over_tolerance_and_not_redone": int(((err32 > MEL_TOL) & ~redone).sum()),
           "smallest_live_band_ratio_among_clips_left_in_float32": worst_ratio, "kFloorRatio": 1e-5,
           "per_family": {str(k): {"n": int((kinds == k).sum()), "auto_max_err_dB": float(err[kinds == k].max()),
                                   "f32_max_err_dB": float(err32[kinds == k].max()), "redone": int(redone[kinds == k].sum())} for k in range(6)}}
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "auto_random.json"), "w") as f:
            json.dump(rec, f, indent=1)
    assert np.isfinite(auto).all() and auto.min() >= -80.0 and np.all(auto.max(axis=(1, 2, 3)) == 0.0)
    assert err.max() <= MEL_TOL, rec
