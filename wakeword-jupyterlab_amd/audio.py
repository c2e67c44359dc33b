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
`load_audio` / `process_audio_file` read the file with the library's native reader (csrc/ww_files.cpp) and decode, mix down and
resample it on the GPU (kernel K0: scipy.signal.resample_poly's Kaiser design -- NOT librosa's soxr resampler, an absent third-party
library: parity unpinned) -- the same code path as the batched loaders, one file at a time.  PCM / float WAV only.
"""
from __future__ import annotations

import random

import numpy as np
import torch

from . import ops
from .config import AudioConfig, AugmentationConfig, check_audio_config


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
        """= librosa.load(path, sr=16000) (:65-71): the WHOLE file as float32 at 16 kHz mono, or None (prints the error, never raises on a
        bad file).  Native reader, then ONE upload of the file's bytes and ONE K0 launch over all its 1 s windows (descriptors that share the
        byte offset and differ in crop_start; without normalisation K0 computes a window's samples only): linear in the file's length
        (round 3 uploaded and resampled the whole file once per second of it).  No normalisation, no crop."""
        import ctypes as C
        from . import _native as nat
        from .files import CLIP_SAMPLES, DESC_DTYPE
        rd = self.gpu_reader(1)                                   # raises without a GPU: a missing device is not a bad file
        try:
            slot = rd.next_slot()
            try:
                descs, status = rd.read([file_path], slot)
            except nat.NativeError as e:
                if e.code != nat.WW_ENOSPACE:
                    raise
                rd.regrow(1, int(e.needed * 1.25) + 4096)
                slot = rd.next_slot()
                descs, status = rd.read([file_path], slot)
            if status[0] != 1:
                raise ValueError(nat.WAV_STATUS.get(int(status[0]), int(status[0])))
            n_out = int(-(-(int(descs["n_frames"][0]) * int(descs["up"][0])) // max(1, int(descs["down"][0]))))
            if n_out == 0:
                return np.zeros(0, dtype=np.float32)
            n_win = -(-n_out // CLIP_SAMPLES)
            win = np.repeat(np.asarray(descs[:1]), n_win)                        # a copy: same bytes, same filter, one window each
            win["crop_start"] = np.arange(n_win, dtype=np.int64) * CLIP_SAMPLES
            dev = self._dev()
            with torch.cuda.device(dev):
                raw_dev = torch.from_numpy(rd.staging(slot)).to(dev)
                descs_dev = torch.from_numpy(win.view(np.uint8).reshape(n_win, DESC_DTYPE.itemsize)).to(dev)
                out = torch.empty((n_win, CLIP_SAMPLES), device=dev, dtype=torch.float32)
                nat.check(nat.lib.ww_decode_resample(C.c_void_p(raw_dev.data_ptr()), C.c_void_p(descs_dev.data_ptr()), n_win, 0,
                                                     C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                audio = out.reshape(-1)[:n_out].cpu().numpy()
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
        """load -> normalise over the whole file -> random crop / zero pad -> (augment) -> log-mel (:125-138), all on the GPU:
        native reader -> K0 -> (KA) -> K1.  The random draws are python `random`'s in the reference's order (crop, then augmentation)."""
        pcm, ok = self.load_clips_gpu([file_path])
        if not bool(ok[0]):
            return None
        if augment:
            pcm = self.augment_batch(pcm)
        return self.mel_batch(pcm, normalize=False)[0, 0].cpu().numpy()

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
