"""CPU checks of the augmentation oracle (oracle/augment_oracle.py) and of the host-side tables it shares with the library.

Parity against librosa itself is UNPINNED (librosa / soxr are not installed and the reference holds no fixtures for
augment_audio, wakeword_training_script.py:103-123): the restatement is checked through properties of the published
algorithms and against scipy where scipy implements the same thing (scipy.signal.stft/istft with librosa's framing)."""
import random

import numpy as np
import pytest
from scipy import signal

import wakeword_jupyterlab_amd as pkg
from oracle import augment_oracle as ao


def _clip(i=3):
    return pkg.synth.make_clip(i)


def test_stft_matches_scipy_framing():
    y = _clip()
    D = ao.stft(y)
    assert D.shape == (1025, 32) and D.dtype == np.complex64
    # scipy.signal.stft with boundary='zeros' pads n_fft/2 zeros like librosa center=True/'constant'; it scales by 1/sum(w)
    _, _, Z = signal.stft(y.astype(np.float64), window=ao.hann(), nperseg=2048, noverlap=1536, boundary="zeros", padded=False)
    Z = Z[:, :32] * ao.hann().sum()
    assert np.abs(D - Z).max() <= 2e-4 * np.abs(Z).max()


def test_istft_inverts_stft():
    y = _clip(5)
    z = ao.istft(ao.stft(y), len(y))
    assert z.dtype == np.float32 and np.abs(z - y).max() < 2e-6


def test_phase_vocoder_rate_one_is_identity_and_lengths():
    y = _clip(7)
    D = ao.stft(y)
    S = ao.phase_vocoder(D, 1.0)
    # rate 1: magnitudes are reproduced exactly, phases up to the float32 accumulator's rounding
    assert np.abs(np.abs(S) - np.abs(D)).max() <= 1e-6 * np.abs(D).max()
    for rate in (0.7, 0.8414, 1.0, 1.19, 1.3):
        assert ao.phase_vocoder(D, rate).shape[1] == int(np.ceil(32 / rate))
        assert len(ao.time_stretch(y, rate)) == int(round(16000 / rate))


def test_phase_vocoder_conditioning_sets_the_gpu_tolerance():
    """librosa's float32 phase accumulator makes time_stretch sensitive to float32-level noise in the STFT: perturbing D by
    2e-7 of its peak (what a float32 FFT does) moves the OUTPUT by ~1e-4 of its rms.  The GPU tests' tolerances
    (tests/test_gpu_augment.py: rms 1e-3, max 3e-3 of the peak per vocoder pass) are a small multiple of this floor."""
    rng = np.random.default_rng(0)
    y = _clip(3) / np.abs(_clip(3)).max()
    D = ao.stft(y)
    base = ao.istft(ao.phase_vocoder(D, 0.8), 20000).astype(np.float64)
    Dp = (D + 2e-7 * np.abs(D).max() * (rng.standard_normal(D.shape) + 1j * rng.standard_normal(D.shape))).astype(np.complex64)
    pert = ao.istft(ao.phase_vocoder(Dp, 0.8), 20000).astype(np.float64)
    rel = np.sqrt(((pert - base) ** 2).mean()) / np.sqrt((base ** 2).mean())
    assert 3e-5 < rel < 1e-3


def test_pitch_shift_moves_a_tone():
    t = np.arange(16000) / 16000.0
    tone = np.sin(2 * np.pi * 440.0 * t).astype(np.float32)
    for n_steps in (-3.0, 2.0, 12.0 * np.log2(1.19)):
        z = ao.pitch_shift(tone, n_steps)
        assert z.shape == (16000,) and z.dtype == np.float32
        peak = np.abs(np.fft.rfft(z * np.hanning(16000))).argmax()
        assert abs(peak - 440.0 * 2 ** (n_steps / 12.0)) <= 2.0


def test_resampler_table_matches_the_library_and_interpolates_a_sine():
    import ctypes as C
    from wakeword_jupyterlab_amd import _native as nat
    tab = np.zeros(32769, np.float32)
    assert nat.lib.ww_kaiser_best_host(tab.ctypes.data) == 0
    ref = ao.kaiser_best_table()
    assert np.abs(tab - ref).max() <= 1e-7 and tab[0] == np.float32(ao.KB_ROLLOFF)
    t = np.arange(8000) / 16000.0
    x = np.sin(2 * np.pi * 500.0 * t)
    for ratio in (0.84, 1.0, 1.19):
        y = ao.resample(x, ratio)
        assert len(y) == int(np.ceil(8000 * ratio))
        want = np.sin(2 * np.pi * 500.0 * np.arange(len(y)) / (16000.0 * ratio))
        core = slice(200, len(y) - 200)                      # away from the edges the windowed sinc is exact to ~1e-5
        assert np.abs(y[core] - want[core]).max() < 2e-4


def test_noise_generator_is_the_builds_hash_rng():
    assert np.array_equal(ao.hash_normal(77, 1000), pkg.synth.normal(77, 1000))
    n = ao.hash_normal(5, 16000)
    assert abs(n.mean()) < 0.03 and abs(n.std() - 1.0) < 0.03


def test_draw_plan_follows_the_reference_order_of_draws():
    rng = random.Random(1234)
    plans = [ao.draw_plan(rng) for _ in range(400)]
    on = lambda k, off: np.mean([p[k] != off for p in plans])                               # noqa: E731
    assert 0.7 < on("n_steps", None) < 0.9 and 0.7 < on("rate", None) < 0.9 and 0.7 < on("sigma", 0.0) < 0.9
    assert all(abs(p["shift"]) <= 4800 for p in plans)
    assert all(p["rate"] is None or 0.7 <= p["rate"] <= 1.3 for p in plans)
    assert all(p["crop"] == 0 or p["crop"] <= int(round(16000 / p["rate"])) - 16000 for p in plans)
    # AudioProcessor.draw_augment_plan consumes python's `random` stream exactly like the oracle's restatement
    from wakeword_jupyterlab_amd.audio import AudioProcessor
    proc = AudioProcessor.__new__(AudioProcessor)
    proc.config = pkg.config.AudioConfig
    random.seed(99)
    got = [proc.draw_augment_plan() for _ in range(20)]
    rng = random.Random(99)
    assert got == [ao.draw_plan(rng) for _ in range(20)]


def test_augment_composes_in_reference_order():
    y = _clip(11)
    plan = {"shift": -1234, "n_steps": None, "rate": None, "crop": 0, "sigma": 0.0, "seed": 0}
    assert np.array_equal(ao.augment(y, plan), np.roll(y, -1234))
    plan = {"shift": 0, "n_steps": None, "rate": 0.8, "crop": 100, "sigma": 0.0, "seed": 0}
    z = ao.augment(y, plan)
    assert np.array_equal(z, ao.time_stretch(y, 0.8)[100:16100])
    plan = {"shift": 0, "n_steps": None, "rate": 1.25, "crop": 0, "sigma": 0.15, "seed": 9}
    z = ao.augment(y, plan)
    base = np.pad(ao.time_stretch(y, 1.25), (0, 16000 - 12800))
    assert np.allclose(z, base + 0.15 * ao.hash_normal(9, 16000), atol=1e-6)
