"""CPU restatement of the load -> normalise -> crop/pad front of process_audio_file (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/wakeword_training_script.py:65-83,125-133:
    librosa.load(path, sr=16000)  = soundfile decode to float32 (ints / 2^(bits-1)), librosa.to_mono (channel mean),
                                    resample to 16 kHz
    normalize_audio               = x / max|x| over the whole file
    pad_or_truncate               = random crop (start drawn by the caller) or right zero-pad to 16000

PARITY UNPINNED for the resampler: librosa 0.10's default is soxr_hq (a third-party library, not installed).  This
restatement -- and the GPU kernel -- use scipy.signal.resample_poly's Kaiser-windowed-sinc polyphase design instead;
files already at 16 kHz are decoded exactly.
"""
from math import gcd

import numpy as np
from scipy.signal import resample_poly

SAMPLE_RATE = 16000
CLIP = 16000


def decode(samples: np.ndarray, sample_rate: int) -> np.ndarray:
    """samples: float [n_frames, channels] already scaled to [-1, 1) as soundfile does -> mono 16 kHz float64."""
    x = np.asarray(samples, dtype=np.float32)
    mono = x.mean(axis=1, dtype=np.float32) if x.shape[1] > 1 else x[:, 0]
    if sample_rate == SAMPLE_RATE:
        return mono.astype(np.float64)
    g = gcd(SAMPLE_RATE, int(sample_rate))
    return resample_poly(mono.astype(np.float64), SAMPLE_RATE // g, int(sample_rate) // g)


def load_normalise_crop(samples, sample_rate, crop_start=0, normalize=True) -> np.ndarray:
    y = decode(samples, sample_rate)
    if normalize and len(y):
        with np.errstate(invalid="ignore", divide="ignore"):
            y = y / np.max(np.abs(y))
    y = y[crop_start:crop_start + CLIP]
    return np.pad(y, (0, CLIP - len(y))).astype(np.float32)
