"""Streaming config (BASELINE configs[4]): per-hop sliding 1 s window over many microphones, hipGraph replayed.

Per-window semantics = predict_wakeword (wakeword_training.ipynb cell 19): normalise, log-mel, forward, softmax.
"""
import numpy as np
import pytest
import torch

import wakeword_jupyterlab_amd as pkg
from oracle import mel_oracle, model_oracle

pytestmark = pytest.mark.gpu


def _softmax1(logits):
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    return (e / e.sum(axis=1, keepdims=True))[:, 1]


@pytest.mark.parametrize("hop", [160, 400])
def test_streaming_matches_windowed_oracle(hop):
    dev = torch.device("cuda", 0)
    n_mics, n_hops = 5, 16000 // hop + 37
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = pkg.SimpleWakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    # 5 microphones, each a different synthetic stream longer than the window
    total = n_hops * hop
    streams = np.stack([np.concatenate([pkg.synth.make_clip(10 * i + j) for j in range(total // 16000 + 1)])[:total]
                        for i in range(n_mics)]).astype(np.float32)
    det = pkg.StreamingDetector(m, n_mics=n_mics, hop_samples=hop, threshold=0.5)
    checked = 0
    for k in range(n_hops):
        chunk = torch.from_numpy(streams[:, k * hop:(k + 1) * hop]).to(dev)
        prob = det.step(chunk)
        done = (k + 1) * hop
        if done >= 16000 and (k % 11 == 0 or k == n_hops - 1):
            det.stream.synchronize()
            win = streams[:, done - 16000:done]
            assert np.array_equal(det.window().cpu().numpy(), win)            # ring holds exactly the last second
            ref_logits = model_oracle.forward_np(mel_oracle.logmel_batch(win, normalize=True), sd)
            assert np.abs(det.logits.cpu().numpy() - ref_logits).max() <= 1e-3
            assert np.abs(prob.cpu().numpy() - _softmax1(ref_logits)).max() <= 1e-3
            assert np.array_equal(det.detections().cpu().numpy(), prob.cpu().numpy() >= 0.5)
            checked += 1
    assert checked >= 3
    det.close()


def test_streaming_partial_window_is_zero_padded_on_the_left_and_silence_is_nan():
    dev = torch.device("cuda", 0)
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = pkg.SimpleWakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    det = pkg.StreamingDetector(m, n_mics=2, hop_samples=160)
    x = pkg.synth.make_clips(40, 2)
    x[1] = 0.0                                             # a dead microphone
    for k in range(30):
        det.step(torch.from_numpy(x[:, k * 160:(k + 1) * 160]).to(dev))
    det.stream.synchronize()
    win = np.zeros((2, 16000), np.float32)
    win[:, 16000 - 4800:] = x[:, :4800]                    # history not yet filled: zeros are the oldest samples
    assert np.array_equal(det.window().cpu().numpy(), win)
    ref = model_oracle.forward_np(mel_oracle.logmel_batch(win[:1], normalize=True), sd)
    assert np.abs(det.logits.cpu().numpy()[0] - ref[0]).max() <= 1e-3
    # silent window: x / max|x| = 0/0 in the reference (normalize_audio) -> NaN probability, never a detection
    assert np.isnan(det.prob.cpu().numpy()[1]) and not bool(det.detections()[1])
    det.close()


def test_streaming_clean_tone_goes_through_the_float64_path():
    """A microphone carrying a noise-free tone: its windows sit on the float FFT's rounding floor, so the captured per-hop graph's
    second log-mel launch (auto mode: the float64 kernel in ring mode, marked clips only) must redo them -- logits against the
    windowed oracle like any other microphone."""
    dev = torch.device("cuda", 0)
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = pkg.SimpleWakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    hop, n_hops = 400, 55
    t = np.arange(n_hops * hop) / 16000.0
    streams = np.stack([0.5 * np.sin(2 * np.pi * 440.0 * t), pkg.synth.make_clips(3, 2).reshape(-1)[: n_hops * hop],
                        0.3 * np.sin(2 * np.pi * 1234.5 * t) * (t > 0.4)]).astype(np.float32)
    det = pkg.StreamingDetector(m, n_mics=3, hop_samples=hop)
    for k in range(n_hops):
        det.step(torch.from_numpy(streams[:, k * hop:(k + 1) * hop]).to(dev))
        if k in (20, 39, 54):
            det.stream.synchronize()
            done = (k + 1) * hop
            win = np.zeros((3, 16000), np.float32)
            seg = streams[:, max(0, done - 16000):done]
            win[:, 16000 - seg.shape[1]:] = seg
            ref_mel = mel_oracle.logmel_batch(win, normalize=True)
            ref = model_oracle.forward_np(ref_mel, sd)
            assert np.abs(det.logits.cpu().numpy() - ref).max() <= 1e-3
    det.close()


def test_streamer_rejects_bad_hops():
    from wakeword_jupyterlab_amd import _native as nat
    dev = torch.device("cuda", 0)
    m = pkg.SimpleWakewordModel().to(dev).eval()
    for hop in (0, 6, 170, 32000):                          # not a multiple of 4 / does not divide 16000
        with pytest.raises(nat.NativeError):
            pkg.StreamingDetector(m, n_mics=2, hop_samples=hop)
