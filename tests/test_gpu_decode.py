"""K0: GPU decode + mono + resample + whole-file peak normalise + crop/pad, against oracle/decode_oracle.py.

The resampler's reference is scipy.signal.resample_poly (librosa's soxr_hq is not installed: parity with it is unpinned).
"""
import os
import random
import struct

import numpy as np
import pytest
import torch

import wakeword_jupyterlab_amd as pkg
from oracle import decode_oracle, mel_oracle
from wakeword_jupyterlab_amd.audio import AudioProcessor
from wavio import _read_wav

pytestmark = pytest.mark.gpu


def _write_wav(path, x, sr=16000, bits=16, channels=1, fmt=1):
    x = np.asarray(x, dtype=np.float64).reshape(-1, channels)
    if fmt == 3 and bits == 64:
        raw = x.astype("<f8").tobytes()
    elif fmt == 3:
        raw, bits = x.astype("<f4").tobytes(), 32
    elif bits == 16:
        raw = np.clip(np.round(x * 32767), -32768, 32767).astype("<i2").tobytes()
    elif bits == 8:
        raw = np.clip(np.round(x * 127 + 128), 0, 255).astype(np.uint8).tobytes()
    elif bits == 24:
        v = np.clip(np.round(x * 8388607), -8388608, 8388607).astype(np.int32).reshape(-1)
        raw = b"".join(int(t).to_bytes(3, "little", signed=True) for t in v)
    elif bits == 32:
        raw = np.clip(np.round(x * 2147483647), -2147483648, 2147483647).astype("<i4").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, fmt, channels, sr, sr * channels * bits // 8, channels * bits // 8, bits) + b"data" + struct.pack("<I", len(raw))
    with open(path, "wb") as f:
        f.write(hdr + raw)


def _tone(n, sr, seed):
    t = np.arange(n) / sr
    return 0.4 * np.sin(2 * np.pi * (180 + 37 * seed) * t) + 0.2 * np.sin(2 * np.pi * 1234.5 * t) + 0.05 * pkg.synth.normal(seed, n)


CASES = [  # (name, sr, seconds, channels, bits, fmt)
    ("s16_16k", 16000, 1.0, 1, 16, 1), ("s16_16k_short", 16000, 0.4, 1, 16, 1), ("s16_16k_long", 16000, 2.3, 1, 16, 1),
    ("f32_16k_stereo", 16000, 1.0, 2, 32, 3), ("u8_16k", 16000, 0.7, 1, 8, 1), ("s24_16k", 16000, 1.0, 1, 24, 1),
    ("s32_16k", 16000, 0.9, 1, 32, 1), ("s16_44k", 44100, 1.3, 1, 16, 1), ("s16_48k_stereo", 48000, 0.8, 2, 16, 1),
    ("s16_8k", 8000, 1.0, 1, 16, 1), ("s16_22k_long", 22050, 1.9, 1, 16, 1), ("f64_16k", 16000, 0.6, 1, 64, 3), ("f64_32k_stereo", 32000, 1.1, 2, 64, 3),
    # 11.025 kHz: up / down = 640 / 441, a 12,801-tap filter that does not fit LDS (decode_resample_kernel's direct form); 96 kHz: down = 6
    ("s16_11k", 11025, 1.2, 1, 16, 1), ("s16_96k", 96000, 0.5, 1, 16, 1), ("s24_24k_stereo", 24000, 1.6, 2, 24, 1),
]


def test_gpu_decode_matches_oracle(tmp_path):
    proc = AudioProcessor()
    paths = []
    for i, (name, sr, secs, ch, bits, fmt) in enumerate(CASES):
        x = _tone(int(sr * secs), sr, i)
        x = np.stack([x, 0.5 * x[::-1]], 1) if ch == 2 else x
        p = os.path.join(tmp_path, name + ".wav")
        _write_wav(p, x * 0.8, sr, bits, ch, fmt)
        paths.append(p)
    bad = os.path.join(tmp_path, "bad.wav")
    open(bad, "wb").write(b"RIFF....WAVEjunk")
    paths.insert(3, bad)

    for normalize in (True, False):
        random.seed(7)
        out, ok = proc.load_clips_gpu(paths, normalize=normalize)
        out = out.cpu().numpy()
        assert out.shape == (len(paths), 16000) and list(ok) == [True] * 3 + [False] + [True] * (len(paths) - 4)
        assert not out[3].any()
        random.seed(7)                                            # replay the random crops the loader drew
        for i, p in enumerate(paths):
            if not ok[i]:
                continue
            samples, sr = _read_wav(p)
            n_out = len(decode_oracle.decode(samples, sr))
            start = random.randint(0, n_out - 16000) if n_out > 16000 else 0
            ref = decode_oracle.load_normalise_crop(samples, sr, start, normalize)
            err = np.abs(out[i] - ref).max()
            assert err <= (2e-7 if sr == 16000 else 5e-6), (p, err)   # exact conversion at 16 kHz; f32 filter sums otherwise
            if normalize:
                assert abs(np.abs(ref).max() - 1.0) < 1e-6 or n_out > 16000


def test_dataset_batches_with_gpu_decode_feed_the_model(tmp_path):
    # end to end through the reference-shaped objects: files -> K0 -> K1 -> model, against the CPU oracles
    from oracle import model_oracle
    dev = torch.device("cuda", 0)
    files = []
    for i in range(5):
        p = os.path.join(tmp_path, f"c{i}.wav")
        _write_wav(p, pkg.synth.make_clip(i)[: 16000 - 1500 * i] * 0.5, 16000)
        files.append(p)
    ds = pkg.WakewordDataset(files[:2], files[2:], AudioProcessor(), verbose=False)
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = pkg.SimpleWakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    got_mel, got_logits, got_t = [], [], []
    with torch.no_grad():
        for data, target in ds.batches(batch_size=3):
            got_mel.append(data.cpu().numpy()); got_logits.append(m(data).cpu().numpy()); got_t.append(target.cpu().numpy())
    mel = np.concatenate(got_mel); logits = np.concatenate(got_logits)
    assert mel.shape == (5, 1, 80, 32) and np.concatenate(got_t).reshape(-1).tolist() == [1, 1, 0, 0, 0]
    ref_pcm = np.stack([decode_oracle.load_normalise_crop(*_read_wav(p)) for p in files])
    ref_mel = mel_oracle.logmel_batch(ref_pcm, normalize=False)
    assert np.abs(mel - ref_mel).max() <= 1e-4
    assert np.abs(logits - model_oracle.forward_np(ref_mel, sd)).max() <= 1e-3
    # per-item access (reference signature) agrees with the batched path
    item, label = ds[1]
    assert item.shape == (1, 80, 32) and label.tolist() == [1] and np.abs(item.numpy() - mel[1]).max() <= 1e-4


def test_reader_slots_overlap_without_corrupting_batches(tmp_path):
    """The file-fed pipeline's buffer rotation: eight batches of different files go through a 2-slot reader back to back with no
    synchronisation in between (host threads fill slot s+1 while the upload / K0 of slot s are in flight; a slot's next upload
    waits for the K0 that read its device twin, its next read for the upload that left its staging).  Every batch must come
    out as the oracle decodes it -- a protocol slip shows up as another batch's samples."""
    from wakeword_jupyterlab_amd.files import WavBatchReader
    dev = torch.device("cuda", 0)
    n_batches, per = 8, 48
    paths, refs = [], []
    for i in range(n_batches * per):
        x = pkg.synth.make_clip(1000 + i)[: 16000 - 37 * (i % 11)] * (0.3 + 0.01 * (i % 50))
        p = os.path.join(tmp_path, f"p{i:04d}.wav")
        _write_wav(p, x, 16000)
        paths.append(p)
        refs.append(decode_oracle.load_normalise_crop(*_read_wav(p)))
    rd = WavBatchReader(max_clips=per, max_raw_bytes=per * 32016, threads=4, slots=2, device=dev)
    outs = [torch.empty((per, 16000), device=dev) for _ in range(n_batches)]
    for b in range(n_batches):
        _, ok = rd.load(paths[b * per:(b + 1) * per], normalize=True, out=outs[b])
        assert ok.all()
    torch.cuda.synchronize()
    got = torch.cat(outs).cpu().numpy()
    assert np.abs(got - np.stack(refs)).max() <= 2e-7
    # a batch that does not fit: load() re-creates the reader with the size the library asked for
    long = os.path.join(tmp_path, "long.wav")
    _write_wav(long, _tone(16000 * 120, 16000, 3) * 0.5, 16000)                 # 3.8 MB > the 1.5 MB staging
    random.seed(3)
    out, ok = rd.load([long, paths[0]], normalize=True)
    random.seed(3)
    start = random.randint(0, 16000 * 119)
    assert ok.all() and np.abs(out[0].cpu().numpy() - decode_oracle.load_normalise_crop(*_read_wav(long), start)).max() <= 2e-7
    assert np.abs(out[1].cpu().numpy() - refs[0]).max() <= 2e-7
    rd.close()


def test_per_file_api_is_the_reader_and_k0(tmp_path, capsys):
    """load_audio (:65-71) and process_audio_file (:125-138), one file at a time: native reader -> K0 (-> K1), no host decode.
    load_audio returns the WHOLE file at 16 kHz (one K0 window per second of output)."""
    x = pkg.synth.make_clip(3) * 0.5
    p16 = os.path.join(tmp_path, "a.wav"); _write_wav(p16, x)
    pf = os.path.join(tmp_path, "f.wav"); _write_wav(pf, x, fmt=3)
    p8 = os.path.join(tmp_path, "u8.wav"); _write_wav(p8, x, bits=8)
    pst = os.path.join(tmp_path, "st.wav"); _write_wav(pst, np.stack([x, -x * 0.5], 1), channels=2)
    long = os.path.join(tmp_path, "long44.wav"); _write_wav(long, _tone(int(44100 * 2.3), 44100, 5) * 0.7, sr=44100)
    short = os.path.join(tmp_path, "short.wav"); _write_wav(short, x[:5000])
    proc = AudioProcessor()
    a = proc.load_audio(p16)
    assert a.dtype == np.float32 and a.shape == (16000,) and np.abs(a - x).max() < 1 / 32768 + 1e-7
    assert np.array_equal(proc.load_audio(pf), x.astype(np.float32))
    assert np.abs(proc.load_audio(p8) - x).max() < 1 / 64
    assert np.abs(proc.load_audio(pst) - 0.25 * x).max() < 1e-4              # mono = channel mean, as librosa.load
    assert proc.load_audio(short).shape == (5000,)
    samples, sr = _read_wav(long)
    want = decode_oracle.decode(samples, sr)
    got = proc.load_audio(long)
    assert got.shape == want.shape and len(got) > 2 * 16000 and np.abs(got - want).max() <= 5e-6      # three K0 windows, stitched
    # failure: print + None, never raise (reference :66-71)
    bad = os.path.join(tmp_path, "bad.wav"); open(bad, "wb").write(b"not a wav")
    assert proc.load_audio(bad) is None and proc.load_audio(os.path.join(tmp_path, "missing.wav")) is None
    assert proc.process_audio_file(bad) is None
    assert "Error loading" in capsys.readouterr().out
    # process_audio_file = the batched path with one file, bit for bit; the crop is python random's draw, as pad_or_truncate's
    m = proc.process_audio_file(p16)
    pcm, ok = proc.load_clips_gpu([p16])
    assert m.shape == (80, 32) and np.array_equal(m, proc.mel_batch(pcm, normalize=False)[0, 0].cpu().numpy())
    random.seed(5)
    m_long = proc.process_audio_file(long)
    random.seed(5)
    start = random.randint(0, len(want) - 16000)
    ref = proc.audio_to_mel(decode_oracle.load_normalise_crop(samples, sr, start))
    assert np.abs(m_long - ref).max() <= 1e-3                                  # dB; the clips differ by the resampler's float32 sums
    random.seed(11)
    m_aug = proc.process_audio_file(p16, augment=True)
    assert m_aug.shape == (80, 32) and np.isfinite(m_aug).all() and not np.array_equal(m_aug, m)


def test_two_loaders_over_one_processor_and_seeded_augmented_epochs(tmp_path):
    """The reference hands ONE AudioProcessor to its train, validation and test datasets (:448-458).  (1) Two loaders over it iterated at
    the same time (validating in the middle of an epoch, zip(train, val)) each get their own native reader and serve the batches they serve
    alone; (2) an augmented epoch over files LONGER than 1 s under one `random.seed` repeats bit for bit: the crop draws and the augmentation
    draws come from python `random` on one thread, in a fixed order (round 3 drew the crops on the reader's helper thread)."""
    proc = AudioProcessor()
    pos, neg = [], []
    for i in range(14):
        p = os.path.join(tmp_path, f"f{i:02d}.wav")
        n = 16000 + 700 * i if i % 2 else 16000 - 900 * i                      # half of them need pad_or_truncate's random crop
        _write_wav(p, np.resize(pkg.synth.make_clip(i), n) * 0.6, 16000)
        (pos if i < 6 else neg).append(p)
    train = pkg.WakewordDataset(pos, neg, proc, augment=True, verbose=False)
    val = pkg.WakewordDataset(pos[0:6:2], neg[0:8:2], proc, verbose=False)                # the files of at most 1 s: no crop draw, no augmentation

    def epoch(ds, seed):
        random.seed(seed)
        return [(d.cpu().numpy(), t.cpu().numpy()) for d, t in ds.loader(batch_size=4)]
    a, b, c = epoch(train, 21), epoch(train, 21), epoch(train, 22)
    assert len(a) == 4 and all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, b))
    assert any(not np.array_equal(x[0], y[0]) for x, y in zip(a, c))
    alone = [d.cpu().numpy() for d, _ in val.loader(batch_size=4)]            # deterministic
    random.seed(5)
    together = []
    for (dt, _), (dv, _) in zip(train.loader(batch_size=4), val.loader(batch_size=4)):
        assert torch.isfinite(dt).all()
        together.append(dv.cpu().numpy())
    assert len(together) == 2 and all(np.array_equal(x, y) for x, y in zip(together, alone))


def test_load_audio_of_a_long_recording_is_one_upload_and_one_launch(tmp_path):
    """load_audio (:65-71) returns the whole file; a 12 s recording is twelve 1 s windows of ONE K0 launch (descriptors sharing the bytes),
    at 16 kHz bit-exact, at 48 kHz against the resampling oracle like the short files."""
    proc = AudioProcessor()
    x = np.resize(pkg.synth.make_clip(9) * 0.4, 16000 * 12 + 4321)
    p = os.path.join(tmp_path, "long16.wav"); _write_wav(p, x, 16000)
    got = proc.load_audio(p)
    samples, sr = _read_wav(p)
    assert got.shape == (len(x),) and np.array_equal(got, decode_oracle.decode(samples, sr))
    p48 = os.path.join(tmp_path, "long48.wav"); _write_wav(p48, _tone(int(48000 * 7.7), 48000, 3) * 0.7, sr=48000)
    samples, sr = _read_wav(p48)
    want = decode_oracle.decode(samples, sr)
    got = proc.load_audio(p48)
    assert got.shape == want.shape and len(got) > 7 * 16000 and np.abs(got - want).max() <= 5e-6
