"""WakewordDataset drop-in (reference: /root/reference/wakeword_training_script.py:187-216, notebook cell 9).

`dataset[idx]` returns `(FloatTensor [1, 80, 32], LongTensor [1])` exactly like the reference.  Because a
GPU-backed `__getitem__` must not initialise HIP inside forked DataLoader workers, the per-item call is
meant for `num_workers=0`; the fast path is `batches(batch_size)`, which decodes on the host and runs the
log-mel kernel once per batch, yielding device tensors with `default_collate`'s shapes
(`data [B,1,80,32]`, `target [B,1]`).

One deliberate deviation: when a file fails to load the reference substitutes `np.zeros((80, 31))`
(:210-211), whose width (31) differs from real items (32) and makes `default_collate` raise.  Here the
substitute is zeros of the real width, [80, 32].

`augment=True` (the training split, :456) runs AudioProcessor.augment_audio's transforms on the GPU (kernels KA)
between decode and log-mel: per item in `__getitem__`, per batch in `batches()`.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

from .config import N_FRAMES


class WakewordDataset(Dataset):
    def __init__(self, wakeword_files, negative_files, processor, augment=False, verbose=True):
        self.wakeword_files = list(wakeword_files)
        self.negative_files = list(negative_files)
        self.processor = processor
        self.augment = augment
        self.files = self.wakeword_files + self.negative_files
        self.labels = [1] * len(self.wakeword_files) + [0] * len(self.negative_files)
        self.unreadable = 0        # items served as zeros because the file could not be decoded (non-WAV, corrupt, missing)
        if verbose:
            print(f"Dataset created with {len(self.files)} samples")
            print(f"Wakeword samples: {len(self.wakeword_files)}")
            print(f"Negative samples: {len(self.negative_files)}")

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        if torch.utils.data.get_worker_info() is not None:
            # the reference's own call site is DataLoader(..., num_workers=2) (:461-463); a forked worker cannot
            # initialise HIP, and a spawned one would build its own context per worker -- refuse instead of failing obscurely
            raise RuntimeError("WakewordDataset.__getitem__ runs the log-mel kernel on the GPU and cannot be called from "
                               "DataLoader worker processes: use DataLoader(..., num_workers=0) or dataset.batches(batch_size)")
        mel_spec = self.processor.process_audio_file(self.files[idx], augment=self.augment)
        if mel_spec is None:
            self.unreadable += 1
            mel_spec = np.zeros((self.processor.config.N_MELS, N_FRAMES))
        return torch.FloatTensor(np.asarray(mel_spec, dtype=np.float32)).unsqueeze(0), torch.LongTensor([self.labels[idx]])

    def batches(self, batch_size=16, gpu_decode=True):
        """Yield (data [B,1,80,32] on the GPU, target [B,1] on the GPU) in file order.
        gpu_decode=True: the host only reads bytes; decode/resample/normalise/crop run in kernel K0, log-mel in K1."""
        dev = self.processor._dev()
        if gpu_decode and getattr(self, "_encoded", None) is None:
            from .files import EncodedPaths
            self._encoded = EncodedPaths(self.files)          # the file list as the native reader takes it, converted once
        for s in range(0, len(self.files), batch_size):
            paths = self.files[s:s + batch_size]
            if gpu_decode:
                pcm_dev, ok = self.processor.load_clips_gpu(self._encoded, normalize=True, lo=s, hi=min(len(self.files), s + batch_size))
            else:
                pcm, ok = self.processor.load_clips(paths)
                pcm_dev = torch.from_numpy(pcm).to(dev)
            # both loaders already peak-normalised each file before the crop/pad, as the reference does (:131-133)
            if self.augment:
                pcm_dev = self.processor.augment_batch(pcm_dev)          # process_audio_file :134-135
            data = self.processor.mel_batch(pcm_dev, normalize=False)
            if not ok.all():
                self.unreadable += int((~ok).sum())
                print(f"WakewordDataset: {int((~ok).sum())} unreadable file(s) in this batch served as zeros "
                      f"({self.unreadable} so far), e.g. {paths[int(np.argmin(ok))]}")
                data[torch.from_numpy(~ok).to(dev)] = 0.0
            target = torch.tensor(self.labels[s:s + batch_size], dtype=torch.long, device=dev).unsqueeze(1)
            yield data, target
