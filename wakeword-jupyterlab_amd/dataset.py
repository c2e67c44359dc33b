"""WakewordDataset drop-in (reference: /root/reference/wakeword_training_script.py:187-216, notebook cell 9).

`dataset[idx]` returns `(FloatTensor [1, 80, 32], LongTensor [1])` exactly like the reference.  Because a
GPU-backed `__getitem__` must not initialise HIP inside forked DataLoader workers, the per-item call is
meant for `num_workers=0`; the fast path is `loader(batch_size, shuffle)` (and `batches(batch_size)`, the same in file
order): the library's reader threads feed kernels K0 / KA / K1 once per batch, yielding device tensors with
`default_collate`'s shapes (`data [B,1,80,32]`, `target [B,1]`) -- the drop-in for the reference's
`DataLoader(dataset, batch_size=16, shuffle=True, num_workers=2)` line.

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

    def batches(self, batch_size=16):
        """Yield (data [B,1,80,32] on the GPU, target [B,1] on the GPU) in file order: the host only reads bytes;
        decode / resample / normalise / crop run in kernel K0, log-mel in K1."""
        return iter(GpuBatchLoader(self, batch_size, shuffle=False))

    def loader(self, batch_size=16, shuffle=False, drop_last=False):
        """What `DataLoader(dataset, batch_size=.., shuffle=.., num_workers=2)` is to the reference's loops
        (/root/reference/wakeword_training_script.py:461-463), without worker processes: an iterable with `len()`, re-iterable
        (a new permutation per epoch when shuffle=True, drawn from torch's generator like RandomSampler: `torch.manual_seed`
        repeats it), yielding `(data [B,1,80,32], target [B,1])` on the GPU.  Files are read by the library's reader threads into
        pinned staging, decoded (K0), augmented when the dataset says so (KA) and turned into log-mel (K1) one batch at a time."""
        return GpuBatchLoader(self, batch_size, shuffle=shuffle, drop_last=drop_last)


class GpuBatchLoader:
    def __init__(self, dataset, batch_size=16, shuffle=False, drop_last=False):
        if batch_size < 1:
            raise ValueError("batch_size must be positive")
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), bool(shuffle), bool(drop_last)
        self._encoded = None

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def order(self):
        """Index order of one epoch (a fresh permutation from torch's default generator when shuffling)."""
        n = len(self.dataset)
        idx = torch.randperm(n).tolist() if self.shuffle else list(range(n))
        return idx[: len(self) * self.batch_size] if self.drop_last else idx

    def __iter__(self):
        ds = self.dataset
        dev = ds.processor._dev()
        idx = self.order()
        files = [ds.files[i] for i in idx]
        labels = [ds.labels[i] for i in idx]
        from .files import EncodedPaths
        if not self.shuffle:
            if self._encoded is None:
                self._encoded = EncodedPaths(files)           # the file list as the native reader takes it, converted once
            enc = self._encoded
        else:
            enc = EncodedPaths(files)                         # this epoch's order

        def decoded():
            """(first index, pcm on the device, ok mask) per batch: the native reader one batch ahead (files.WavBatchReader.stream)."""
            from . import _native as nat
            reader = ds.processor.gpu_reader(self.batch_size)
            if reader._streaming:
                # the processor's reader is feeding another loader right now (the reference hands ONE processor to its train, validation and
                # test datasets, :448-458: validating in the middle of an epoch, zip(train, val)): this iteration gets a reader of its own
                from .files import WavBatchReader
                reader = WavBatchReader(max_clips=max(64, self.batch_size), max_raw_bytes=max(64, self.batch_size) * 65536, slots=3,
                                        device=reader.device)
            s = 0
            while s < len(files):
                try:
                    for pcm_dev, ok in reader.stream(enc, self.batch_size, normalize=True, verbose=False, start=s):
                        yield s, pcm_dev, ok
                        s += self.batch_size
                except nat.NativeError as e:                   # a batch of long files: larger staging, then on from that batch
                    if e.code != nat.WW_ENOSPACE:
                        raise
                    reader.regrow(self.batch_size, int(e.needed * 1.25) + 4096)

        for s, pcm_dev, ok in decoded():
            e = min(len(files), s + self.batch_size)
            # both loaders already peak-normalised each file before the crop/pad, as the reference does (:131-133)
            if ds.augment:
                pcm_dev = ds.processor.augment_batch(pcm_dev)          # process_audio_file :134-135
            data = ds.processor.mel_batch(pcm_dev, normalize=False)
            if not ok.all():
                ds.unreadable += int((~ok).sum())
                print(f"WakewordDataset: {int((~ok).sum())} unreadable file(s) in this batch served as zeros "
                      f"({ds.unreadable} so far), e.g. {files[s + int(np.argmin(ok))]}")
                data[torch.from_numpy(~ok).to(dev)] = 0.0
            target = torch.tensor(labels[s:e], dtype=torch.long, device=dev).unsqueeze(1)
            yield data, target


def DataLoader(dataset, batch_size=1, shuffle=False, *args, num_workers=0, drop_last=False, **kwargs):
    """Drop-in for the name the reference imports (`from torch.utils.data import Dataset, DataLoader`,
    /root/reference/wakeword_training_script.py:9): with it the script's loader lines (:461-463,
    `DataLoader(train_dataset, batch_size=16, shuffle=True, num_workers=2)`) run unchanged.  For a WakewordDataset it returns
    `dataset.loader(batch_size, shuffle, drop_last)` -- `num_workers` is accepted and not used: the library's reader threads and the GPU
    kernels do what the worker processes did (a forked worker could not touch the GPU anyway).  Any other dataset goes to torch's
    DataLoader untouched.  A sampler / batch_sampler / collate_fn on a WakewordDataset is refused (the batch is built on the GPU)."""
    if isinstance(dataset, WakewordDataset):
        unsupported = [k for k in ("sampler", "batch_sampler", "collate_fn") if kwargs.get(k) is not None]
        if args or unsupported:
            raise NotImplementedError("DataLoader(WakewordDataset, ...): positional extras / " + ", ".join(unsupported or ["sampler"]) +
                                      " are not supported -- the batch is assembled on the GPU (dataset.loader)")
        if batch_size is None:
            raise NotImplementedError("DataLoader(WakewordDataset, batch_size=None): unbatched loading is dataset[i]")
        return dataset.loader(batch_size=batch_size, shuffle=bool(shuffle), drop_last=bool(drop_last))
    return torch.utils.data.DataLoader(dataset, batch_size, shuffle, *args, num_workers=num_workers, drop_last=drop_last, **kwargs)
