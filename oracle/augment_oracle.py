"""CPU restatement of AudioProcessor.augment_audio (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/wakeword_training_script.py:103-123:
    np.roll(audio, shift)                                   time shift
    librosa.effects.pitch_shift(y, sr=16000, n_steps=n)     = resample(time_stretch(y, rate=2^(-n/12)), sr/rate -> sr), fix_length
    librosa.effects.time_stretch(y, rate=r)                 = istft(phase_vocoder(stft(y), r), length=round(len/r))
    pad_or_truncate(.., len(audio))                         random crop (start drawn by the caller) or right zero-pad
    audio + np.random.normal(0, 0.15, len)                  additive noise
each applied with probability 0.8; the random draws are the caller's (a plan, see `draw_plan`).

PARITY UNPINNED.  The arithmetic is librosa 0.10.1's (README.md:386), a third-party dependency that is not installed
here, and the reference holds no fixtures for it.  This file restates the published algorithms:
  * stft: n_fft 2048, hop 512, periodic Hann, center=True, pad_mode='constant', float64 rFFT stored as complex64;
  * phase_vocoder: linear magnitude interpolation, phase advance accumulated IN FLOAT32 (librosa's accumulator is
    np.angle(D[..., 0]) of a complex64 matrix, updated in place), phasor in float32;
  * istft: window * irfft, overlap-add in frame order in float32, division by the window sum-square where it
    exceeds float32 tiny, centre trimming, `length` semantics of librosa 0.10.1;
  * resample: librosa's default res_type 'soxr_hq' is another absent third-party library; as in K0 a windowed-sinc
    interpolator stands in -- resampy's published 'kaiser_best' design (64 zero crossings, 512 table entries per
    crossing with linear interpolation, Kaiser beta 14.769656459379492, roll-off 0.9475937167399596), evaluated at
    t = i / ratio;
  * noise: numpy's Mersenne-Twister stream cannot be reproduced on a GPU; both sides use the build's counter-based
    hash generator (wakeword-jupyterlab_amd/synth.py: splitmix64 -> Box-Muller), seed drawn by the caller.
"""
import math
import random

import numpy as np
from scipy.signal import get_window
from scipy.signal.windows import kaiser

N_FFT = 2048
HOP = 512
CLIP = 16000
SR = 16000

_MASK64 = (1 << 64) - 1


# ---- noise: the build's counter-based generator (same integers as wakeword-jupyterlab_amd/synth.py) ------------------
def _mix64(x):
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _hash_u32(seed, stream, n):
    with np.errstate(over="ignore"):
        key = _mix64(np.array([(int(seed) * 0x9E3779B97F4A7C15 + int(stream) * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D)
                               & _MASK64], dtype=np.uint64))[0]
        idx = np.arange(1, n + 1, dtype=np.uint64)
        return (_mix64(key + idx * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)).astype(np.uint32)


def hash_normal(seed, n):
    u1 = (_hash_u32(seed, 1, n).astype(np.float64) + 0.5) / 4294967296.0
    u2 = (_hash_u32(seed, 2, n).astype(np.float64) + 0.5) / 4294967296.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# ---- librosa.stft / phase_vocoder / istft ------------------------------------------------------------------------------
def hann():
    return get_window("hann", N_FFT, fftbins=True)          # float64, periodic


def stft(y):
    """librosa.stft(y) with its defaults -> complex64 [1025, 1 + len(y)//512]."""
    y = np.asarray(y, dtype=np.float32)
    yp = np.pad(y, (N_FFT // 2, N_FFT // 2), mode="constant")
    n_frames = 1 + (len(yp) - N_FFT) // HOP
    w = hann()
    out = np.empty((N_FFT // 2 + 1, n_frames), dtype=np.complex64)
    for t in range(n_frames):
        out[:, t] = np.fft.rfft(w * yp[t * HOP:t * HOP + N_FFT])
    return out


def phase_vocoder(D, rate):
    """librosa.phase_vocoder(D, rate=rate) for complex64 D (hop_length = n_fft // 4)."""
    rate = float(rate)
    n_bins, n_in = D.shape
    time_steps = np.arange(0, n_in, rate, dtype=np.float64)
    out = np.zeros((n_bins, len(time_steps)), dtype=np.complex64)
    phi_advance = HOP * (np.arange(n_bins, dtype=np.float64) * (2.0 * np.pi) / N_FFT)     # fft_frequencies(sr=2 pi)
    phase_acc = np.angle(D[:, 0])                               # float32 (!): librosa accumulates in this array
    assert phase_acc.dtype == np.float32
    Dp = np.pad(D, ((0, 0), (0, 2)), mode="constant")
    for t, step in enumerate(time_steps):
        c0, c1 = Dp[:, int(step)], Dp[:, int(step) + 1]
        alpha = np.float32(np.mod(step, 1.0))                    # numpy 1.24: a float64 scalar does not upcast a float32 array
        mag = np.float32(1.0 - np.mod(step, 1.0)) * np.abs(c0) + alpha * np.abs(c1)
        out[:, t] = (np.cos(phase_acc) + 1j * np.sin(phase_acc)).astype(np.complex64) * mag     # util.phasor, complex64
        dphase = (np.angle(c1) - np.angle(c0)) - phi_advance    # float32 difference, then float64
        dphase = dphase - 2.0 * np.pi * np.round(dphase / (2.0 * np.pi))
        phase_acc[:] = (phase_acc.astype(np.float64) + (phi_advance + dphase)).astype(np.float32)   # in-place f32 += f64
    return out


def istft(S, length):
    """librosa.istft(S, length=length) (center=True, hann, hop 512) -> float32 [length]."""
    n_frames = min(S.shape[1], int(math.ceil((length + N_FFT) / HOP)))
    w = hann()
    full = np.zeros(N_FFT + HOP * (n_frames - 1), dtype=np.float32)
    wss = np.zeros_like(full)
    wsq = (w * w).astype(np.float32)
    for t in range(n_frames):
        ytmp = (w * np.fft.irfft(S[:, t].astype(np.complex128), n=N_FFT)).astype(np.float32)
        full[t * HOP:t * HOP + N_FFT] += ytmp
        wss[t * HOP:t * HOP + N_FFT] += wsq
    y = np.zeros(length, dtype=np.float32)
    seg = full[N_FFT // 2:N_FFT // 2 + length]
    y[:len(seg)] = seg
    ws = np.zeros(length, dtype=np.float32)
    seg = wss[N_FFT // 2:N_FFT // 2 + length]
    ws[:len(seg)] = seg
    ok = ws > np.finfo(np.float32).tiny
    y[ok] /= ws[ok]
    return y


def time_stretch(y, rate):
    """librosa.effects.time_stretch(y, rate=rate)."""
    return istft(phase_vocoder(stft(y), rate), int(round(len(y) / float(rate))))


# ---- resampler stand-in (resampy 'kaiser_best') ----------------------------------------------------------------------
KB_ZEROS, KB_PRECISION = 64, 9
KB_BETA, KB_ROLLOFF = 14.769656459379492, 0.9475937167399596


def kaiser_best_table():
    """Half of the interpolation window, resampy.filters.sinc_window(64, 9, kaiser(beta), rolloff): 32769 float64."""
    n = (1 << KB_PRECISION) * KB_ZEROS
    sinc_win = KB_ROLLOFF * np.sinc(KB_ROLLOFF * np.linspace(0, KB_ZEROS, num=n + 1, endpoint=True))
    taper = kaiser(2 * n + 1, KB_BETA)[n:]
    return taper * sinc_win


def resample(x, ratio):
    """n_out = ceil(len * ratio) samples of x evaluated at t = i / ratio (resampy.resample_f with 'kaiser_best')."""
    x = np.asarray(x, dtype=np.float64)
    win = kaiser_best_table()
    num_table = 1 << KB_PRECISION
    scale = min(1.0, ratio)
    if ratio < 1.0:
        win = win * ratio
    delta = np.append(np.diff(win), 0.0)
    index_step = int(scale * num_table)
    n_out = int(math.ceil(len(x) * ratio))
    nwin, n_orig = len(win), len(x)
    y = np.zeros(n_out)
    for t in range(n_out):
        time_register = t * (1.0 / ratio)
        n = int(time_register)
        if n >= n_orig:
            continue
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        idx = offset + index_step * np.arange(i_max)
        y[t] += np.dot(win[idx] + eta * delta[idx], x[n - np.arange(i_max)])
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        idx = offset + index_step * np.arange(k_max)
        y[t] += np.dot(win[idx] + eta * delta[idx], x[n + 1 + np.arange(k_max)])
    return y.astype(np.float32)


def pitch_rate(n_steps):
    return 2.0 ** (-float(n_steps) / 12.0)


def pitch_shift(y, n_steps):
    """librosa.effects.pitch_shift(y, sr=16000, n_steps=n_steps) with the resampler stand-in."""
    rate = pitch_rate(n_steps)
    ratio = float(SR) / (float(SR) / rate)                     # target_sr / orig_sr as librosa forms it
    z = resample(time_stretch(y, rate), ratio)
    out = np.zeros(len(y), dtype=np.float32)
    n = min(len(y), len(z))
    out[:n] = z[:n]
    return out


# ---- the plan: one record per clip -----------------------------------------------------------------------------------
def draw_plan(rng: random.Random, prob=0.8, noise=0.15, shift_max=0.3, pitch_max=3.0, speed=(0.7, 1.3), n=CLIP):
    """The random draws of augment_audio in the reference's order (random.random / random.uniform / random.randint)."""
    plan = {"shift": 0, "n_steps": None, "rate": None, "crop": 0, "sigma": 0.0, "seed": 0}
    if rng.random() < prob:
        plan["shift"] = int(rng.uniform(-shift_max, shift_max) * SR)
    if rng.random() < prob:
        plan["n_steps"] = rng.uniform(-pitch_max, pitch_max)
    if rng.random() < prob:
        plan["rate"] = rng.uniform(*speed)
        stretched = int(round(n / plan["rate"]))
        if stretched > n:
            plan["crop"] = rng.randint(0, stretched - n)       # pad_or_truncate's random crop
    if rng.random() < prob:
        plan["sigma"] = noise
        plan["seed"] = rng.getrandbits(32)
    return plan


def augment(audio, plan):
    y = np.asarray(audio, dtype=np.float32).copy()
    n = len(y)
    if plan["shift"]:
        y = np.roll(y, plan["shift"])
    if plan["n_steps"] is not None:
        y = pitch_shift(y, plan["n_steps"])
    if plan["rate"] is not None:
        z = time_stretch(y, plan["rate"])
        z = z[plan["crop"]:plan["crop"] + n] if len(z) > n else z
        y = np.pad(z, (0, n - len(z))).astype(np.float32)
    if plan["sigma"]:
        y = (y.astype(np.float64) + plan["sigma"] * hash_normal(plan["seed"], n)).astype(np.float32)
    return y
