"""CPU restatement of the reference log-mel front-end (TEST INFRASTRUCTURE ONLY).

Follows, call for call, what
    /root/reference/wakeword_training_script.py:73-101
does through librosa 0.10.1 (pinned in /root/reference/README.md:386; numpy
1.24.3 at README.md:388):

    normalize_audio   :73-76   x / max|x|
    pad_or_truncate   :78-83   right zero-pad / random crop
    audio_to_mel      :85-101  librosa.feature.melspectrogram(sr=16000, n_mels=80,
                               n_fft=2048, hop_length=512, win_length=2048,
                               fmin=0, fmax=8000)  ->  librosa.power_to_db(ref=np.max)

PARITY UNPINNED: librosa is a third-party dependency that is absent from
/root/reference and from this image, and the reference ships no golden vector
for this path.  The algorithm below is librosa 0.10.x's published one
(stft -> |.|**2 -> filters.mel (Slaney) -> power_to_db), restated with the same
dtype at every step ("librosa32" functions).  `*_f64` variants do everything in
double and exist only to budget the float32 error.

numpy 2.x scalar-promotion differs from the pinned numpy 1.24; every dtype is
therefore explicit below rather than left to promotion rules.
"""
from __future__ import annotations

import numpy as np
from scipy.signal import get_window

# AudioConfig, /root/reference/wakeword_training_script.py:29-37
SAMPLE_RATE = 16000
DURATION = 1.0
N_MELS = 80
N_FFT = 2048
HOP_LENGTH = 512
WIN_LENGTH = 2048
FMIN = 0.0
FMAX = 8000.0
CLIP_SAMPLES = int(SAMPLE_RATE * DURATION)          # 16000
N_FRAMES = 1 + CLIP_SAMPLES // HOP_LENGTH           # 32 (center=True)
N_BINS = 1 + N_FFT // 2                             # 1025
AMIN = 1e-10                                        # librosa.power_to_db default
TOP_DB = 80.0                                       # librosa.power_to_db default


# --------------------------------------------------------------------------
# librosa.filters.mel (htk=False, norm='slaney', dtype=float32)
# --------------------------------------------------------------------------
def hz_to_mel(f):
    """librosa.core.convert.hz_to_mel, htk=False (Slaney / Auditory Toolbox)."""
    f = np.asanyarray(f, dtype=np.float64)
    f_min, f_sp = 0.0, 200.0 / 3
    mels = (f - f_min) / f_sp
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        log_t = f >= min_log_hz
        mels[log_t] = min_log_mel + np.log(f[log_t] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def mel_to_hz(m):
    """librosa.core.convert.mel_to_hz, htk=False."""
    m = np.asanyarray(m, dtype=np.float64)
    f_min, f_sp = 0.0, 200.0 / 3
    freqs = f_min + f_sp * m
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if m.ndim:
        log_t = m >= min_log_mel
        freqs[log_t] = min_log_hz * np.exp(logstep * (m[log_t] - min_log_mel))
    elif m >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (m - min_log_mel))
    return freqs


def mel_edges(n_mels=N_MELS, fmin=FMIN, fmax=FMAX):
    """librosa.mel_frequencies(n_mels + 2, fmin, fmax, htk=False): the 82 band edges in Hz."""
    mels = np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2)
    return mel_to_hz(mels)


def mel_filterbank(sr=SAMPLE_RATE, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX):
    """librosa.filters.mel(..., htk=False, norm='slaney', dtype=np.float32) -> [n_mels, 1+n_fft//2] f32.

    The triangle is rounded to float32 when stored, then scaled by the float64
    Slaney area normaliser and rounded again (the in-place `weights *= enorm`).
    """
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_edges(n_mels, fmin, fmax)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights = (weights.astype(np.float64) * enorm[:, np.newaxis]).astype(np.float32)
    return weights


# --------------------------------------------------------------------------
# librosa.stft(center=True, pad_mode='constant', window='hann')
# --------------------------------------------------------------------------
def hann_window(n=WIN_LENGTH):
    """scipy.signal.get_window('hann', n, fftbins=True), float64 (what librosa.filters.get_window returns)."""
    return get_window("hann", n, fftbins=True)


def frame_signal(y, n_fft=N_FFT, hop=HOP_LENGTH):
    """Centered framing: pad n_fft//2 zeros each side, frame t = yp[hop*t : hop*t + n_fft] -> [n_fft, T]."""
    yp = np.pad(np.asarray(y), (n_fft // 2, n_fft // 2), mode="constant")
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    return yp[idx]


def stft_librosa32(y):
    """librosa.stft on float32 input: float64 window * frame, numpy rfft in double, stored as complex64."""
    y = np.asarray(y, dtype=np.float32)
    frames = frame_signal(y)                                   # f32 [2048, T]
    win = hann_window()[:, None]                               # f64
    spec = np.fft.rfft(win * frames.astype(np.float64), axis=0)  # complex128 [1025, T]
    return spec.astype(np.complex64)


def melspectrogram_librosa32(y, mel_basis=None):
    """librosa.feature.melspectrogram(power=2.0) -> float32 [80, T]."""
    if mel_basis is None:
        mel_basis = mel_filterbank()
    S = np.abs(stft_librosa32(y))                              # float32
    S = S * S                                                  # `** 2.0` on float32 stays float32
    assert S.dtype == np.float32
    return np.matmul(mel_basis, S).astype(np.float32)          # einsum('ft,mf->mt') in float32


def power_to_db_librosa32(S):
    """librosa.power_to_db(S, ref=np.max, amin=1e-10, top_db=80.0) on a float32 array."""
    S = np.asarray(S, dtype=np.float32)
    log_spec = np.float32(10.0) * np.log10(np.maximum(np.float32(AMIN), S))
    # ref_value is a numpy scalar: under numpy 1.24 `np.maximum(amin, np.float32)` and the
    # log10 run in float64, then the in-place subtract rounds the scalar to float32.
    ref_value = np.float64(np.max(S))
    ref_db = np.float32(10.0 * np.log10(np.maximum(AMIN, ref_value)))
    log_spec = log_spec - ref_db
    log_spec = np.maximum(log_spec, log_spec.max() - np.float32(TOP_DB))
    assert log_spec.dtype == np.float32
    return log_spec


def audio_to_mel(audio, mel_basis=None):
    """AudioProcessor.audio_to_mel, /root/reference/wakeword_training_script.py:85-101."""
    audio = np.asarray(audio)
    if len(audio) == 0:
        return np.zeros((N_MELS, int(SAMPLE_RATE * DURATION / HOP_LENGTH) + 1))
    return power_to_db_librosa32(melspectrogram_librosa32(audio, mel_basis))


def normalize_audio(audio):
    """AudioProcessor.normalize_audio, :73-76 (silent clip -> 0/0 = NaN, as in the reference)."""
    audio = np.asarray(audio)
    if len(audio) == 0:
        return audio
    with np.errstate(invalid="ignore", divide="ignore"):
        return audio / np.max(np.abs(audio))


def pad_or_truncate(audio, target_length=CLIP_SAMPLES, rng=None):
    """AudioProcessor.pad_or_truncate, :78-83.  `rng` is a `random.Random`-like object (randint)."""
    audio = np.asarray(audio)
    if len(audio) > target_length:
        import random as _random
        r = rng if rng is not None else _random
        start = r.randint(0, len(audio) - target_length)
        return audio[start : start + target_length]
    return np.pad(audio, (0, target_length - len(audio)), mode="constant")


def process_clip(audio, normalize=True, mel_basis=None):
    """process_audio_file minus the file decode (:125-138, augment=False): normalise -> pad -> log-mel."""
    audio = np.asarray(audio, dtype=np.float32)
    if normalize:
        audio = normalize_audio(audio)
    audio = pad_or_truncate(audio)
    return audio_to_mel(audio.astype(np.float32), mel_basis)


def logmel_batch(pcm, normalize=True):
    """[B, n] float32 -> [B, 1, 80, 32] float32, one clip at a time as WakewordDataset.__getitem__ does (:204-216)."""
    basis = mel_filterbank()
    out = np.empty((len(pcm), 1, N_MELS, N_FRAMES), dtype=np.float32)
    for i, clip in enumerate(pcm):
        out[i, 0] = process_clip(clip, normalize, basis)
    return out


# --------------------------------------------------------------------------
# float64 end to end: error-budget reference, not the parity target
# --------------------------------------------------------------------------
def logmel_f64(y, normalize=True):
    y = np.asarray(y, dtype=np.float64)
    if normalize:
        with np.errstate(invalid="ignore", divide="ignore"):
            y = y / np.max(np.abs(y))
    y = np.pad(y, (0, max(0, CLIP_SAMPLES - len(y))))
    frames = frame_signal(y)
    spec = np.fft.rfft(hann_window()[:, None] * frames, axis=0)
    S = spec.real ** 2 + spec.imag ** 2
    mel = mel_filterbank().astype(np.float64) @ S
    db = 10.0 * np.log10(np.maximum(AMIN, mel)) - 10.0 * np.log10(np.maximum(AMIN, mel.max()))
    return np.maximum(db, db.max() - TOP_DB)
