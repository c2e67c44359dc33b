"""KA: AudioProcessor.augment_audio on the GPU against oracle/augment_oracle.py.

PARITY UNPINNED against librosa / soxr / numpy's RNG themselves (none can be run here; the reference holds no fixtures):
the oracle restates librosa 0.10.1's phase vocoder and stands in for the resampler and the noise stream (see its header).
Tolerances: time shift exact; noise 1e-6; the phase-vocoder paths are float32 FFTs against the oracle's float64 ones
and carry librosa's float32 phase accumulator (values up to ~7e4 rad, one ulp = 0.008 rad), so a last-bit difference in an
atan2 or in a weak STFT bin moves that bin's phase by an ulp.  tests/test_oracle_augment.py measures that floor on the
oracle itself (float32-level noise in D -> ~1e-4..3e-4 of the output's rms); the bounds here are a small multiple of it:
max |err| <= 3e-3 of the clip's peak and rms err <= 1e-3 of its rms per vocoder pass (twice that for pitch + stretch).
"""
import random

import numpy as np
import pytest
import torch

import wakeword_jupyterlab_amd as pkg
from oracle import augment_oracle as ao
from wakeword_jupyterlab_amd import _native as nat
from wakeword_jupyterlab_amd import ops
from wakeword_jupyterlab_amd.audio import AudioProcessor

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
OFF = {"shift": 0, "n_steps": None, "rate": None, "crop": 0, "sigma": 0.0, "seed": 0}


def _clips(n, start=0):
    x = pkg.synth.make_clips(start, n)
    return x / np.abs(x).max(axis=1, keepdims=True)          # process_audio_file normalises before augmenting (:131-135)


def _run(x, plans):
    return ops.augment(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(DEV), plans).cpu().numpy()


def _check(got, want, max_rel=3e-3, rms_rel=1e-3):
    assert got.shape == want.shape and got.dtype == np.float32
    err = got.astype(np.float64) - want.astype(np.float64)
    peak, rms = np.abs(want).max(), np.sqrt((want.astype(np.float64) ** 2).mean())
    assert np.abs(err).max() <= max_rel * peak, (np.abs(err).max() / peak)
    assert np.sqrt((err ** 2).mean()) <= rms_rel * rms, (np.sqrt((err ** 2).mean()) / rms)


def test_identity_plan_copies():
    x = _clips(3)
    assert np.array_equal(_run(x, [OFF] * 3), x)


def test_time_shift_is_np_roll_exactly():
    x = _clips(5)
    shifts = [1, -1, 4799, -4800, 16000 + 7]
    got = _run(x, [dict(OFF, shift=s) for s in shifts])
    for i, s in enumerate(shifts):
        assert np.array_equal(got[i], np.roll(x[i], s))


def test_noise_matches_the_hash_generator():
    x = _clips(2)
    got = _run(x, [dict(OFF, sigma=0.15, seed=12345), dict(OFF, sigma=0.15, seed=2 ** 32 - 1)])
    for i, seed in enumerate((12345, 2 ** 32 - 1)):
        want = (x[i].astype(np.float64) + 0.15 * ao.hash_normal(seed, 16000)).astype(np.float32)
        assert np.abs(got[i] - want).max() <= 1e-6
    assert abs((got[0] - x[0]).std() - 0.15) < 0.01


@pytest.mark.parametrize("rate", [0.7, 0.83, 1.0, 1.19, 1.3])
def test_time_stretch_against_the_oracle(rate):
    x = _clips(2, start=3)                                    # clip 3 is tonal, clip 4 noise
    n_str = int(round(16000 / rate))
    crop = (n_str - 16000) // 3 if n_str > 16000 else 0
    got = _run(x, [dict(OFF, rate=rate, crop=crop)] * 2)
    for i in range(2):
        z = ao.time_stretch(x[i], rate)
        want = z[crop:crop + 16000] if len(z) > 16000 else np.pad(z, (0, 16000 - len(z)))
        _check(got[i], want.astype(np.float32))
        if n_str < 16000:
            assert np.all(got[i][n_str:] == 0.0)              # pad_or_truncate's zero pad


@pytest.mark.parametrize("n_steps", [-3.0, -1.3, 0.5, 3.0])
def test_pitch_shift_against_the_oracle(n_steps):
    x = _clips(2, start=6)
    got = _run(x, [dict(OFF, n_steps=n_steps)] * 2)
    for i in range(2):
        _check(got[i], ao.pitch_shift(x[i], n_steps))


def test_pitch_shift_moves_a_tone_on_the_gpu():
    t = np.arange(16000) / 16000.0
    tone = np.sin(2 * np.pi * 440.0 * t).astype(np.float32)[None]
    z = _run(tone, [dict(OFF, n_steps=3.0)])[0]
    peak = np.abs(np.fft.rfft(z * np.hanning(16000))).argmax()
    assert abs(peak - 440.0 * 2 ** (3.0 / 12.0)) <= 2.0


def test_full_plans_mixed_batch():
    rng = random.Random(2024)
    n = 12
    x = _clips(n, start=20)
    plans = [ao.draw_plan(rng) for _ in range(n)]
    plans[0] = dict(OFF)                                      # all stages off for one clip, only some on for others
    plans[1] = dict(plans[1], n_steps=None)
    plans[2] = dict(plans[2], rate=None, crop=0)
    got = _run(x, plans)
    for i in range(n):
        _check(got[i], ao.augment(x[i], plans[i]), max_rel=6e-3, rms_rel=2e-3)    # two vocoder passes in sequence


def test_two_call_form_is_graph_capturable_and_bitwise_equal():
    """ww_augment_plans_prepare (host arithmetic) + ww_augment_records_f32 (kernels only): captured ONCE into a graph together with
    the host -> device copy of the records, replayed with new plans -- every replay equals ww_augment_f32 on the same plans bit for
    bit, including a batch where no clip asks for pitch (the captured graph still launches that stage: copy-through)."""
    import ctypes as C
    n = 10
    x = torch.from_numpy(np.ascontiguousarray(_clips(n, start=40), dtype=np.float32)).to(DEV)
    rb = int(nat.lib.ww_augment_record_bytes())
    rec_host = torch.empty(n * rb, dtype=torch.uint8).pin_memory()
    rec_dev = torch.empty(n * rb, dtype=torch.uint8, device=DEV)
    out = torch.empty_like(x)
    ws = torch.empty(int(nat.lib.ww_augment_workspace_bytes(n)), dtype=torch.uint8, device=DEV)

    def plans_array(plans):
        arr = (nat.AugmentPlan * n)()
        for a, p in zip(arr, plans):
            a.shift, a.crop_start = p["shift"], p["crop"]
            a.pitch_rate = 2.0 ** (-p["n_steps"] / 12.0) if p["n_steps"] is not None else 0.0
            a.stretch_rate = p["rate"] or 0.0
            a.noise_sigma, a.noise_seed = p["sigma"], p["seed"]
        return arr
    rng = random.Random(77)
    batches = [[ao.draw_plan(rng) for _ in range(n)] for _ in range(3)]
    batches.append([dict(p, n_steps=None) for p in batches[0]])                       # nobody asks for pitch
    ops.init()
    nat.check(nat.lib.ww_augment_plans_prepare(C.cast(plans_array(batches[0]), C.c_void_p), n, C.c_void_p(rec_host.data_ptr())))
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                                      # warm-up outside capture (LDS opt-in, tables)
        rec_dev.copy_(rec_host, non_blocking=True)
        nat.check(nat.lib.ww_augment_records_f32(x.data_ptr(), n, 16000, rec_dev.data_ptr(), out.data_ptr(), ws.data_ptr(), C.c_void_p(side.cuda_stream)))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        rec_dev.copy_(rec_host, non_blocking=True)
        nat.check(nat.lib.ww_augment_records_f32(x.data_ptr(), n, 16000, rec_dev.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for plans in batches:
        nat.check(nat.lib.ww_augment_plans_prepare(C.cast(plans_array(plans), C.c_void_p), n, C.c_void_p(rec_host.data_ptr())))
        g.replay()
        torch.cuda.synchronize()
        want = ops.augment(x, plans_array(plans))
        assert torch.equal(out, want)


def test_rejects_bad_plans_before_launching():
    x = torch.zeros((1, 16000), device=DEV)
    for bad, code in ((dict(OFF, rate=0.5), nat.WW_EUNSUPPORTED), (dict(OFF, rate=40.0), nat.WW_EUNSUPPORTED),
                      (dict(OFF, pitch_rate=0.3), nat.WW_EUNSUPPORTED), (dict(OFF, rate=0.8, crop=5000), nat.WW_EINVAL),
                      (dict(OFF, rate=1.2, crop=1), nat.WW_EINVAL), (dict(OFF, sigma=-1.0), nat.WW_EINVAL)):
        with pytest.raises(nat.NativeError) as e:
            ops.augment(x, [bad])
        assert e.value.code == code
    with pytest.raises(ValueError):
        ops.augment(x, [OFF, OFF])
    with pytest.raises(RuntimeError):
        ops.augment(torch.zeros((1, 16000)), [OFF])
    assert ops.augment(torch.zeros((0, 16000), device=DEV), []).shape == (0, 16000)


def test_audio_processor_augment_audio_and_dataset(tmp_path):
    proc = AudioProcessor()
    y = _clips(1)[0]
    random.seed(7)
    z = proc.augment_audio(y)
    assert isinstance(z, np.ndarray) and z.shape == (16000,) and z.dtype == np.float32 and np.isfinite(z).all()
    random.seed(7)
    plan = proc.draw_augment_plan()
    _check(z, ao.augment(y, plan), max_rel=6e-3, rms_rel=2e-3)
    with pytest.raises(ValueError):
        proc.augment_audio(y[:100])
    # WakewordDataset(augment=True): per item and per batch, shapes as the reference's
    import struct
    paths = []
    for i in range(3):
        raw = np.clip(np.round(_clips(1, start=40 + i)[0] * 32767), -32768, 32767).astype("<i2").tobytes()
        hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + \
            b"data" + struct.pack("<I", len(raw))
        p = tmp_path / f"c{i}.wav"
        p.write_bytes(hdr + raw)
        paths.append(str(p))
    ds = pkg.WakewordDataset(paths[:2], paths[2:], proc, augment=True, verbose=False)
    data, target = ds[0]
    assert data.shape == (1, 80, 32) and target.shape == (1,) and float(data.max()) == 0.0
    for data, target in ds.batches(batch_size=2):
        assert data.shape[1:] == (1, 80, 32) and data.is_cuda and torch.isfinite(data).all()
