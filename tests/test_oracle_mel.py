"""Pins for the log-mel restatement.

librosa itself is absent (un-vendored third-party dependency, not installed, no golden vectors in
the reference): parity with librosa is UNPINNED.  What is checked instead:
  * the structural facts of librosa.filters.mel(sr=16000, n_fft=2048, n_mels=80, fmax=8000) recorded
    in SURVEY.md section 8(c);
  * the STFT stage against torch.stft, an independent implementation;
  * the float32-faithful path against an all-float64 evaluation (error budget);
  * the invariants power_to_db(ref=np.max, top_db=80) implies.
"""
import numpy as np
import pytest
import torch

import wakeword_jupyterlab_amd.synth as synth
from oracle import mel_oracle as mo


@pytest.fixture(scope="module")
def clips():
    return synth.make_clips(0, 6)


def test_mel_filterbank_structure():
    M = mo.mel_filterbank()
    nz = M != 0
    assert M.shape == (80, 1025) and M.dtype == np.float32
    assert nz.sum() == 2004
    assert nz.sum(1).min() == 9 and nz.sum(1).max() == 75
    assert nz.sum(0).max() == 2 and not nz[:, 0].any() and not nz[:, 1024].any()
    assert abs(float(M.sum()) - 10.2395) < 1e-3 and abs(float(M.max()) - 0.02667) < 1e-5
    assert list(np.nonzero(nz[0])[0][[0, -1]]) == [1, 9] and list(np.nonzero(nz[79])[0][[0, -1]]) == [949, 1023]
    e = mo.mel_edges()
    assert np.allclose(e[:3], [0.0, 37.239, 74.478], atol=1e-3) and np.allclose(e[-3:], [7408.54, 7698.59, 8000.0], atol=1e-2)
    assert abs(float(mo.hz_to_mel(8000.0)) - 45.2456) < 1e-3
    # each filter's support is one contiguous run of bins, and consecutive filters overlap
    for i in range(80):
        k = np.nonzero(nz[i])[0]
        assert np.array_equal(k, np.arange(k[0], k[-1] + 1))


def test_window_and_framing():
    w = mo.hann_window()
    n = np.arange(2048)
    assert w.dtype == np.float64 and w[0] == 0.0 and abs(w[1024] - 1.0) < 1e-15
    assert np.abs(w - (0.5 - 0.5 * np.cos(2 * np.pi * n / 2048))).max() < 1e-15
    y = np.arange(1, 16001, dtype=np.float32)
    fr = mo.frame_signal(y)
    assert fr.shape == (2048, 32)
    # valid samples in the edge frames: 1024/1536 at the head, 1664/1152 at the tail
    assert [(fr[:, t] != 0).sum() for t in (0, 1, 30, 31)] == [1024, 1536, 1664, 1152]
    assert fr[1024, 0] == 1.0 and fr[0, 2] == 1.0 and fr[0, 3] == 513.0


def test_stft_against_torch_stft(clips):
    for i in (0, 1):
        y = mo.normalize_audio(clips[i]).astype(np.float32)
        ref = mo.stft_librosa32(y)
        S = torch.stft(torch.from_numpy(y).double(), 2048, 512, 2048,
                       window=torch.hann_window(2048, periodic=True, dtype=torch.float64),
                       center=True, pad_mode="constant", return_complex=True).numpy()
        assert ref.shape == (1025, 32) and ref.dtype == np.complex64
        assert np.abs(S - ref).max() <= 2e-7 * np.abs(S).max()          # complex64 storage rounding only


def test_logmel_invariants_and_float32_budget(clips):
    lm = mo.logmel_batch(clips)
    assert lm.shape == (6, 1, 80, 32) and lm.dtype == np.float32
    for i in range(6):
        # max is 0 up to the one float32 rounding of the scalar ref term; floor is max - 80
        assert abs(float(lm[i].max())) <= 4e-6 and lm[i].min() >= lm[i].max() - 80.0
        assert np.abs(mo.logmel_f64(clips[i]) - lm[i, 0]).max() < 2e-5


def test_gain_invariance_and_normalize_flag(clips):
    # log-mel is invariant to input gain (per-clip max subtracted in dB) while mel power > amin
    a = mo.process_clip(clips[0], normalize=True)
    b = mo.process_clip(clips[0] * np.float32(0.37), normalize=True)
    c = mo.process_clip(clips[0], normalize=False)
    assert np.abs(a - b).max() < 1e-5 and np.abs(a - c).max() < 1e-5


def test_silent_clip_is_nan_like_the_reference():
    # normalize_audio divides by max|x| = 0 (wakeword_training_script.py:73-76): NaN mel
    with np.errstate(all="ignore"):
        out = mo.process_clip(np.zeros(16000, np.float32), normalize=True)
    assert out.shape == (80, 32) and np.isnan(out).all()
    # without the normalisation a silent clip sits on the amin clamp: all 0 dB
    flat = mo.process_clip(np.zeros(16000, np.float32), normalize=False)
    assert np.abs(flat).max() < 1e-5 and np.unique(flat).size == 1


def test_short_clip_is_right_padded_and_empty_clip_is_zeros():
    x = synth.make_clip(3)[:9000]
    a = mo.process_clip(x)
    b = mo.process_clip(np.concatenate([x, np.zeros(7000, np.float32)]))
    assert np.array_equal(a, b)
    assert mo.audio_to_mel(np.zeros(0)).shape == (80, 32)


def test_pad_or_truncate_crop_uses_rng():
    import random
    x = np.arange(20000, dtype=np.float32)
    r = random.Random(4)
    start = random.Random(4).randint(0, 4000)
    assert np.array_equal(mo.pad_or_truncate(x, 16000, r), x[start:start + 16000])
