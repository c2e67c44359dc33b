"""Deterministic synthetic clips and random-init weights.

Everything is a pure function of (seed, index) through a 64-bit integer hash, so
the build container, the GPU box and every rank regenerate bit-identical arrays
without shipping them.

Clip recipe = `create_sample_data` of the reference
(/root/reference/wakeword_training_script.py:350-393): "wakeword" clips are
0.1*randn + 0.3*sin(2*pi*200 t) + 0.2*sin(2*pi*400 t), negatives are 0.2*randn,
t = linspace(0, 1, 16000).  Here clip i is tonal when i % 3 == 0 (the reference's
50:100 ratio) with f0 = 200 + 25*(i mod 16) Hz, so i mod 16 == 0 is the reference
recipe exactly.
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
CLIP_SAMPLES = 16000
_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def hash_u32(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n 32-bit words, word j = top half of mix64(mix64(seed, stream) + golden*(offset+j+1))."""
    with np.errstate(over="ignore"):
        key = _mix64(np.array([(int(seed) * 0x9E3779B97F4A7C15 + int(stream) * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D)
                               & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        return (_mix64(key + idx * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)).astype(np.uint32)


def uniform01(seed: int, stream: int, n: int) -> np.ndarray:
    """float64 uniforms in (0, 1): (u32 + 0.5) / 2**32 -- exact in double."""
    return (hash_u32(seed, stream, n).astype(np.float64) + 0.5) * (1.0 / 4294967296.0)


def normal(seed: int, n: int) -> np.ndarray:
    """float64 standard normals by Box-Muller over two hash streams."""
    u1 = uniform01(seed, 1, n)
    u2 = uniform01(seed, 2, n)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def make_clip(i: int, n: int = CLIP_SAMPLES) -> np.ndarray:
    """Clip i of the synthetic set -> float32 [n]."""
    noise = normal(i, n)
    if i % 3 == 0:
        t = np.linspace(0.0, n / SAMPLE_RATE, n)
        f0 = 200.0 + 25.0 * (i % 16)
        x = 0.1 * noise + 0.3 * np.sin(2 * np.pi * f0 * t) + 0.2 * np.sin(2 * np.pi * 2 * f0 * t)
    else:
        x = 0.2 * noise
    return x.astype(np.float32)


def make_clips(start: int, count: int, n: int = CLIP_SAMPLES) -> np.ndarray:
    """Clips [start, start+count) -> float32 [count, n]."""
    out = np.empty((count, n), dtype=np.float32)
    for j in range(count):
        out[j] = make_clip(start + j, n)
    return out


def make_clips_tiled(start: int, count: int, unique: int = 256, n: int = CLIP_SAMPLES) -> np.ndarray:
    """`count` clips built from `unique` distinct ones (clip j = clip j mod unique, re-scaled by a
    per-clip gain so that no two rows are bitwise equal).  Used for the large bench batches, where
    hashing 65 M samples per rank would dominate start-up; parity is always checked on real clips."""
    base = make_clips(start, min(unique, count), n)
    out = np.empty((count, n), dtype=np.float32)
    for j in range(count):
        gain = np.float32(1.0 - 0.5 * ((j // len(base)) % 64) / 64.0)
        out[j] = base[j % len(base)] * gain
    return out


# ---------------------------------------------------------------------------
# random-init weights, PyTorch default scales, reference state_dict key set
# ---------------------------------------------------------------------------
def _u(seed: int, stream: int, shape, bound: float) -> np.ndarray:
    n = int(np.prod(shape))
    return ((uniform01(seed, stream, n) * 2.0 - 1.0) * bound).astype(np.float32).reshape(shape)


def make_state_dict(arch: str = "simple", seed: int = 1234, hidden: int = 256) -> dict:
    """numpy state_dict with the key set / shapes of the reference modules.

    arch 'simple' -> SimpleWakewordModel (/root/reference/wakeword_training/train_wakeword.py:28-36)
    arch 'full'   -> WakewordModel       (/root/reference/wakeword_training_script.py:141-165)
    Scales follow torch defaults: conv/linear U(+-1/sqrt(fan_in)), LSTM U(+-1/sqrt(hidden)).
    """
    chans = {"simple": [1, 32, 64], "full": [1, 32, 64, 128]}[arch]
    sd, s = {}, 0
    for li in range(1, len(chans)):
        cin, cout = chans[li - 1], chans[li]
        b = 1.0 / np.sqrt(cin * 9)
        sd[f"conv{li}.weight"] = _u(seed, 100 + s, (cout, cin, 3, 3), b); s += 1
        sd[f"conv{li}.bias"] = _u(seed, 100 + s, (cout,), b); s += 1
    b = 1.0 / np.sqrt(hidden)
    for layer, nin in enumerate([chans[-1], hidden]):
        sd[f"lstm.weight_ih_l{layer}"] = _u(seed, 100 + s, (4 * hidden, nin), b); s += 1
        sd[f"lstm.weight_hh_l{layer}"] = _u(seed, 100 + s, (4 * hidden, hidden), b); s += 1
        sd[f"lstm.bias_ih_l{layer}"] = _u(seed, 100 + s, (4 * hidden,), b); s += 1
        sd[f"lstm.bias_hh_l{layer}"] = _u(seed, 100 + s, (4 * hidden,), b); s += 1
    sd["fc.weight"] = _u(seed, 100 + s, (2, hidden), b); s += 1
    sd["fc.bias"] = _u(seed, 100 + s, (2,), b); s += 1
    return sd


def write_wav16(path: str, audio: np.ndarray, sr: int = SAMPLE_RATE) -> None:
    """What `soundfile.write(path, float_audio, sr)` leaves on disk for a mono float array: RIFF/WAVE, PCM-16 (libsndfile's default subtype
    for .wav; full scale = 32767, clipped here where libsndfile would wrap)."""
    import struct
    raw = np.clip(np.rint(np.asarray(audio, dtype=np.float64) * 32767.0), -32768, 32767).astype("<i2").tobytes()
    hdr = (b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16) +
           b"data" + struct.pack("<I", len(raw)))
    with open(path, "wb") as f:
        f.write(hdr + raw)


def create_sample_data(root: str = ".") -> None:
    """`create_sample_data()` of the reference (/root/reference/wakeword_training_script.py:350-392): 50 wakeword clips (0.1 randn + 200 Hz +
    400 Hz), 100 negatives (0.2 randn), 20 five-second noise files, 16 kHz mono PCM-16 WAV, same directory and file names, drawn from
    numpy's global generator like the reference (np.random.seed repeats them).  Host-only convenience for the script's `main`."""
    import os
    print("Creating sample data for training...")
    for d in ("wakeword_data", "negative_data", "background_noise"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    sr = SAMPLE_RATE
    for i in range(50):
        audio = np.random.randn(sr) * 0.1
        t = np.linspace(0, 1.0, sr)
        audio += np.sin(2 * np.pi * 200 * t) * 0.3
        audio += np.sin(2 * np.pi * 400 * t) * 0.2
        write_wav16(os.path.join(root, "wakeword_data", f"wakeword_{i:03d}.wav"), audio, sr)
    for i in range(100):
        write_wav16(os.path.join(root, "negative_data", f"negative_{i:03d}.wav"), np.random.randn(sr) * 0.2, sr)
    for i in range(20):
        write_wav16(os.path.join(root, "background_noise", f"noise_{i:03d}.wav"), np.random.randn(5 * sr) * 0.1, sr)
    print("Sample data created successfully!")
    print("   Wakeword samples: 50")
    print("   Negative samples: 100")
    print("   Background noise samples: 20")
