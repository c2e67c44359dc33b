"""AudioProcessor drop-in (reference: /root/reference/wakeword_training_script.py:61-138).

Same method names, arguments and return conventions:
  load_audio(path)            -> float32 ndarray at 16 kHz mono, or None (prints the error; never raises)
  normalize_audio(audio)      -> audio / max|audio|            (empty-array guard only, as the reference)
  pad_or_truncate(audio, n)   -> random crop (python `random`) or right zero-pad
  audio_to_mel(audio)         -> ndarray [80, 32] log-mel dB     <- HIP kernel K1
  process_audio_file(path)    -> ndarray [80, 32] or None

plus the batched form the GPU wants: `mel_batch(pcm[B, n]) -> torch.Tensor [B, 1, 80, 32]` on the device.

  augment_audio(audio)        -> time shift / pitch shift / time stretch / noise, each with probability 0.8
                                 (random draws from python `random` in the reference's order)  <- HIP kernels KA
`load_audio` decodes PCM / float WAV with the standard library and resamples with scipy's polyphase
filter -- NOT librosa's decoder + soxr resampler; moving decode+resample to the GPU is the first
"next" row.
"""
from __future__ import annotations

import random
import wave

import numpy as np
import torch

from . import ops
from .config import AudioConfig, AugmentationConfig, check_audio_config


def _parse_wav(data: bytes):
    """RIFF/WAVE header walk -> (format tag, channels, sample rate, bits, data offset, data length)."""
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, where = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], int.from_bytes(data[pos + 4:pos + 8], "little")
        if cid == b"fmt ":
            body = data[pos + 8:pos + 8 + size]
            tag, ch, sr = int.from_bytes(body[0:2], "little"), int.from_bytes(body[2:4], "little"), int.from_bytes(body[4:8], "little")
            bits = int.from_bytes(body[14:16], "little")
            if tag == 0xFFFE and len(body) >= 26:
                tag = int.from_bytes(body[24:26], "little")
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            where = (pos + 8, min(size, len(data) - pos - 8))
        pos += 8 + size + (size & 1)
    if fmt is None or where is None:
        raise ValueError("missing fmt/data chunk")
    return fmt + where


def _read_wav(path: str):
    """Minimal RIFF/WAVE reader: 8/16/24/32-bit PCM and 32-bit float, any channel count -> (float32 [n, ch], sr)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], int.from_bytes(data[pos + 4:pos + 8], "little")
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr = int.from_bytes(body[0:2], "little"), int.from_bytes(body[2:4], "little"), int.from_bytes(body[4:8], "little")
            bits = int.from_bytes(body[14:16], "little")
            if tag == 0xFFFE and len(body) >= 26:          # WAVE_FORMAT_EXTENSIBLE: real tag in the GUID
                tag = int.from_bytes(body[24:26], "little")
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError("missing fmt/data chunk")
    tag, ch, sr, bits = fmt
    if tag == 1 and bits == 8:
        x = (np.frombuffer(pcm, np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm[: len(pcm) // 2 * 2], "<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm[: len(pcm) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (np.where(v >= 1 << 23, v - (1 << 24), v)).astype(np.float32) / float(1 << 23)
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], "<i4").astype(np.float32) / float(1 << 31)
    elif tag == 3 and bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], "<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(pcm[: len(pcm) // 8 * 8], "<f8").astype(np.float32)
    else:
        raise ValueError(f"unsupported WAV encoding tag={tag} bits={bits}")
    x = x[: len(x) // ch * ch].reshape(-1, ch)
    return x, sr


class AudioProcessor:
    def __init__(self, config=AudioConfig, device=None):
        check_audio_config(config)
        self.config = config
        self.device = torch.device(device) if device is not None else None

    def _dev(self):
        if self.device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("AudioProcessor.audio_to_mel runs on the MI355X only and no GPU is visible (no CPU fallback)")
            self.device = torch.device("cuda", torch.cuda.current_device())
        return self.device

    # ---- reference API --------------------------------------------------------------------------
    def load_audio(self, file_path):
        try:
            x, sr = _read_wav(file_path)
            audio = x.mean(axis=1) if x.shape[1] > 1 else x[:, 0]           # librosa.load mono=True
            if sr != self.config.SAMPLE_RATE:
                from math import gcd
                from scipy.signal import resample_poly
                g = gcd(int(sr), int(self.config.SAMPLE_RATE))
                audio = resample_poly(audio.astype(np.float64), self.config.SAMPLE_RATE // g, sr // g)
            return np.ascontiguousarray(audio, dtype=np.float32)
        except Exception as e:                                             # reference: print and return None (:66-71)
            print(f"Error loading {file_path}: {e}")
            return None

    def normalize_audio(self, audio):
        if len(audio) == 0:
            return audio
        with np.errstate(invalid="ignore", divide="ignore"):
            return audio / np.max(np.abs(audio))

    def pad_or_truncate(self, audio, target_length):
        if len(audio) > target_length:
            start_idx = random.randint(0, len(audio) - target_length)
            return audio[start_idx:start_idx + target_length]
        return np.pad(audio, (0, target_length - len(audio)), mode="constant")

    def audio_to_mel(self, audio):
        """[n <= 16000] samples -> ndarray [80, 32] (dB).  No normalisation here (reference :85-101)."""
        n_frames = int(self.config.SAMPLE_RATE * self.config.DURATION / self.config.HOP_LENGTH) + 1
        if len(audio) == 0:
            return np.zeros((self.config.N_MELS, n_frames))
        pcm = torch.as_tensor(np.ascontiguousarray(audio, dtype=np.float32)).unsqueeze(0).to(self._dev())
        return ops.logmel(pcm, False)[0, 0].cpu().numpy()

    def draw_augment_plan(self, config=AugmentationConfig, length=None):
        """The random draws of augment_audio (:103-123) in the reference's order -> one plan dict.
        pad_or_truncate's crop start (:116-117) is drawn here too, right after the speed factor."""
        sr = self.config.SAMPLE_RATE
        n = length or int(sr * self.config.DURATION)
        plan = {"shift": 0, "n_steps": None, "rate": None, "crop": 0, "sigma": 0.0, "seed": 0}
        if random.random() < config.AUGMENTATION_PROB:
            plan["shift"] = int(random.uniform(-config.TIME_SHIFT_MAX, config.TIME_SHIFT_MAX) * sr)
        if random.random() < config.AUGMENTATION_PROB:
            plan["n_steps"] = random.uniform(-config.PITCH_SHIFT_MAX, config.PITCH_SHIFT_MAX)
        if random.random() < config.AUGMENTATION_PROB:
            plan["rate"] = random.uniform(config.SPEED_CHANGE_MIN, config.SPEED_CHANGE_MAX)
            stretched = int(round(n / plan["rate"]))
            if stretched > n:
                plan["crop"] = random.randint(0, stretched - n)
        if random.random() < config.AUGMENTATION_PROB:
            plan["sigma"] = float(config.NOISE_FACTOR)
            plan["seed"] = random.getrandbits(32)
        return plan

    def augment_batch(self, pcm, plans=None, config=AugmentationConfig) -> torch.Tensor:
        """pcm [B, 16000] (ndarray or tensor) -> augmented device tensor [B, 16000]; one plan per clip (drawn here if None)."""
        t = torch.as_tensor(pcm, dtype=torch.float32)
        if t.device.type != "cuda":
            t = t.to(self._dev(), non_blocking=True)
        if plans is None:
            plans = [self.draw_augment_plan(config, t.shape[1]) for _ in range(t.shape[0])]
        return ops.augment(t, plans)

    def augment_audio(self, audio, config=AugmentationConfig):
        """[16000] samples -> augmented float32 ndarray [16000] (reference :103-123), on the GPU."""
        a = np.ascontiguousarray(audio, dtype=np.float32)
        if a.shape != (int(self.config.SAMPLE_RATE * self.config.DURATION),):
            raise ValueError(f"augment_audio takes exactly one padded clip of 16000 samples, got {a.shape}")
        return self.augment_batch(a[None, :], config=config)[0].cpu().numpy()

    def process_audio_file(self, file_path, augment=False):
        audio = self.load_audio(file_path)
        if audio is None:
            return None
        audio = self.normalize_audio(audio)
        audio = self.pad_or_truncate(audio, int(self.config.SAMPLE_RATE * self.config.DURATION))
        if augment:
            audio = self.augment_audio(audio)
        return self.audio_to_mel(audio)

    # ---- batched form ---------------------------------------------------------------------------
    def mel_batch(self, pcm, normalize: bool = True) -> torch.Tensor:
        """pcm [B, n<=16000] (ndarray or tensor, any device) -> device tensor [B, 1, 80, 32].
        normalize=True folds normalize_audio + zero-pad + audio_to_mel, the order process_audio_file uses."""
        t = torch.as_tensor(pcm, dtype=torch.float32)
        if t.device.type != "cuda":
            t = t.to(self._dev(), non_blocking=True)
        return ops.logmel(t, normalize)

    def load_clips_gpu(self, paths, normalize: bool = True, lo: int = 0, hi=None):
        """Host: the library's reader threads open the files, walk their RIFF headers and read the sample bytes into pinned
        staging (files.WavBatchReader -> ww_read_wav_batch_host).  GPU (kernel K0): sample conversion, mono mix, polyphase
        resample to 16 kHz, whole-file peak normalisation, random crop / zero pad to 1 s -- process_audio_file :125-133
        up to the mel call.  `paths`: a list, or a files.EncodedPaths with a window [lo, hi).  Returns (device tensor [B, 16000],
        ok mask); unreadable files give a zero row, ok False."""
        from .files import EncodedPaths
        hi = len(paths) if hi is None else hi
        if not isinstance(paths, EncodedPaths):
            paths = list(paths)
        return self.gpu_reader(hi - lo).load(paths, normalize, lo=lo, hi=hi)

    def gpu_reader(self, batch_size: int = 64):
        """This processor's native batch reader (3 staging slots, sized for `batch_size` files of up to 2 s of 16 kHz PCM-16; it grows on demand)."""
        from .files import WavBatchReader
        dev = self._dev()
        if getattr(self, "_reader", None) is None or self._reader.device != dev:
            self._reader = WavBatchReader(max_clips=max(64, batch_size), max_raw_bytes=max(64, batch_size) * 65536, slots=3, device=dev)
        elif self._reader.max_clips < batch_size:
            self._reader.regrow(batch_size, batch_size * 65536)
        return self._reader

    def load_clips(self, paths, target_length=None):
        """Decode, peak-normalise and crop/pad a list of files on the host, in the reference's order
        (process_audio_file :125-138) -> (float32 [B, 16000], ok mask).  Feed to mel_batch(normalize=False)."""
        n = target_length or int(self.config.SAMPLE_RATE * self.config.DURATION)
        out = np.zeros((len(paths), n), dtype=np.float32)
        ok = np.zeros(len(paths), dtype=bool)
        for i, p in enumerate(paths):
            a = self.load_audio(p)
            if a is None:
                continue
            out[i] = self.pad_or_truncate(self.normalize_audio(a), n)
            ok[i] = True
        return out, ok
