"""File-fed batches: many WAV files -> normalised 1 s clips [B, 16000] on the GPU.

Host half of the reference's `AudioProcessor.load_audio` (/root/reference/wakeword_training_script.py:65-71) as its
`DataLoader(batch_size=16, num_workers=2)` drives it (:461-463), for a whole batch of paths at once: the library's thread
pool opens the files, walks the RIFF chunks and reads the sample bytes straight into pinned staging
(`ww_read_wav_batch_host`, csrc/ww_files.cpp); the slot goes to the GPU on the reader's copy stream and kernel K0 decodes,
mixes to mono, resamples, peak-normalises and crops / zero-pads (`ww_wav_batch_decode`).  With two or more slots the host
reads batch k+1 while the GPU works on batch k.  Python only hands over the path list and draws `pad_or_truncate`'s
random crop (:78-83) with `random.randint`, like the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import random

import numpy as np
import torch

from . import _native as nat
from .config import CLIP_SAMPLES

# struct ww_clip_desc as a numpy record (include/wakeword_amd.h): lets the crop draw and the status scan run vectorised
DESC_DTYPE = np.dtype([("byte_offset", "<i8"), ("n_frames", "<i8"), ("channels", "<i4"), ("sample_rate", "<i4"), ("format", "<i4"),
                       ("crop_start", "<i4"), ("up", "<i4"), ("down", "<i4"), ("half_len", "<i4"), ("_pad", "<i4"), ("taps_dev", "<u8")])
assert DESC_DTYPE.itemsize == C.sizeof(nat.ClipDesc)


def host_cpu_share() -> int:
    """CPUs this process may use: the smaller of its affinity mask (what `taskset` / a cpuset granted) and its cgroup's CPU quota
    (cpu.max: a GPU box of this pool shows 256 CPUs in the mask and a quota of 16), divided by the ranks that share the node
    (LOCAL_WORLD_SIZE, set by torch.distributed.run: one process per GPU)."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            quota, period = parse(open(path).read())
            if quota != "max" and int(quota) > 0:
                cpus = min(cpus, max(1, -(-int(quota) // int(period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    return max(1, cpus // ranks)


def default_threads() -> int:
    """Host threads of a reader: this rank's CPU share (host_cpu_share) less one when it is eight or more, at most 32, or WW_READER_THREADS.  A reader thread is busy the
    whole time a batch is being read (page cache warm: open / pread / close are CPU work, not waiting), so threads beyond the share
    only spend the cgroup's quota faster and are throttled for the rest of each 100 ms period: on an MI355X box (quota 16 CPUs), one
    stream over 0.5 M files, 12 / 16 / 24 / 32 / 48 threads fed the pipeline at 0.87-0.90 / 0.87-0.96 / 0.84-0.88 / 0.69-0.74 /
    0.58-0.63 M clips/s (`profiles/r04_reader_threads_sustained.txt`; round 3's "32 beats 16" came from runs shorter than one quota
    period)."""
    env = os.environ.get("WW_READER_THREADS")
    if env:
        return max(1, min(256, int(env)))
    share = host_cpu_share()
    # one CPU of a larger share stays with the consumer (its Python, the HIP runtime's threads): 15 and 16 threads feed the pipeline alike
    # on a 16-CPU quota (1.21-1.59 vs 1.27-1.38 M clips/s over four rounds), but 16 had quota periods throttled and 15 none
    return max(1, min(32, share - 1 if share >= 8 else share))


class EncodedPaths:
    """A path list converted ONCE to the `const char* const*` the library takes (the conversion costs ~0.3 us per path: a quarter of
    what the reader threads need per file).  `reader.read(enc, slot, lo, hi)` / `reader.load(enc, lo=.., hi=..)` then pass a window of it
    without copying -- what a dataset that walks the same file list every epoch wants."""

    def __init__(self, paths):
        self.paths = list(paths)
        self._bytes = [os.fsencode(p) for p in self.paths]            # keeps the char buffers alive
        self.array = (C.c_char_p * max(1, len(self._bytes)))(*self._bytes)

    def __len__(self):
        return len(self.paths)

    def window(self, lo: int, hi: int):
        return C.cast(C.byref(self.array, lo * C.sizeof(C.c_char_p)), C.POINTER(C.c_char_p))


class WavBatchReader:
    """`read(paths, slot)` (host, blocking) -> `decode(slot)` (GPU, asynchronous) -> device tensor [B, 16000].

    max_raw_bytes: sample bytes one batch may hold (16-byte aligned per file); grows on demand in `load()`."""

    def __init__(self, max_clips: int = 4096, max_raw_bytes: int | None = None, threads: int | None = None, slots: int = 2, device=None,
                 host_only: bool = False):
        """host_only=True: staging in ordinary memory and no device twin -- `read()` works without a GPU (tests of the reader
        threads and the RIFF walk), `decode()` raises."""
        self.host_only = bool(host_only)
        if not host_only and not torch.cuda.is_available():
            raise RuntimeError("WavBatchReader feeds the GPU decode kernel (K0) and no GPU is visible (no CPU fallback)")
        self.device = None if host_only else (torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()))
        self.max_clips = int(max_clips)
        self.max_raw_bytes = int(max_raw_bytes if max_raw_bytes is not None else max(1 << 20, self.max_clips * 2 * CLIP_SAMPLES + 4096))
        self.threads = int(threads or default_threads())
        self.slots = int(slots)
        self._h = C.c_void_p()
        with self._ctx():
            nat.check(nat.lib.ww_wav_reader_create(self.threads, self.slots, self.max_clips, self.max_raw_bytes, int(self.host_only), C.byref(self._h)))
        self._n = [0] * self.slots
        self._last_descs = [None] * self.slots
        self._next = 0
        self._streaming = False        # a stream() generator owns the slots and the thread pool until it is exhausted or closed

    def _ctx(self):
        import contextlib
        return contextlib.nullcontext() if self.host_only else torch.cuda.device(self.device)

    def staging(self, slot: int = 0) -> np.ndarray:
        """The slot's staging bytes as the last `read` left them (a copy) -- descs['byte_offset'] index into it."""
        p, n = C.c_void_p(), C.c_int64(0)
        nat.check(nat.lib.ww_wav_reader_staging(self._h, slot, C.byref(p), C.byref(n)))
        return np.frombuffer((C.c_char * max(1, n.value)).from_address(p.value), dtype=np.uint8, count=n.value).copy()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            nat.lib.ww_wav_reader_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def next_slot(self) -> int:
        s = self._next
        self._next = (s + 1) % self.slots
        return s

    def read(self, paths, slot: int = 0, lo: int = 0, hi: int | None = None):
        """Host threads read `paths[lo:hi]` into the slot's pinned staging (`paths`: a list of str, or an EncodedPaths).  Returns
        (descs, status): a numpy record view of the slot's descriptors (writable: set descs['crop_start'] before `decode`) and
        int8 status per file (1 = ok, else _native.WAV_STATUS).  Raises NativeError(WW_ENOSPACE) with `.needed` set when the
        batch does not fit."""
        hi = len(paths) if hi is None else hi
        n = hi - lo
        if isinstance(paths, EncodedPaths):
            arr = paths.window(lo, hi)
        else:
            arr = (C.c_char_p * max(1, n))(*[os.fsencode(p) for p in paths[lo:hi]])
        status = np.zeros(max(1, n), dtype=np.int8)
        descs_p = C.POINTER(nat.ClipDesc)()
        need = C.c_int64(0)
        with self._ctx():
            rc = nat.lib.ww_read_wav_batch_host(self._h, arr, n, slot, C.byref(descs_p), status.ctypes.data, C.byref(need))
        if rc == nat.WW_ENOSPACE:
            e = nat.NativeError(rc, (nat.lib.ww_last_error() or b"").decode("utf-8", "replace"))
            e.needed = int(need.value)
            raise e
        nat.check(rc)
        self._n[slot] = n
        self._last_descs[slot] = C.cast(descs_p, C.c_void_p)      # where the library keeps this slot's descriptors (tests)
        if n == 0:
            return np.zeros(0, dtype=DESC_DTYPE), status[:0]
        buf = (C.c_char * (n * DESC_DTYPE.itemsize)).from_address(C.addressof(descs_p.contents))
        return np.frombuffer(buf, dtype=DESC_DTYPE, count=n), status[:n]

    def decode(self, slot: int = 0, normalize: bool = True, out: torch.Tensor | None = None) -> torch.Tensor:
        """Upload the slot and run K0 on torch's current stream -> [n, 16000] float32 on the device (asynchronous)."""
        n = self._n[slot]
        if self.host_only:
            raise RuntimeError("decode: this reader is host_only (no device twin)")
        if out is None:
            out = torch.empty((n, CLIP_SAMPLES), device=self.device, dtype=torch.float32)
        elif out.shape != (n, CLIP_SAMPLES) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError(f"decode: out must be a contiguous float32 [{n}, {CLIP_SAMPLES}] tensor on {self.device}")
        if self.host_only:
            raise RuntimeError("decode: this reader is host_only (no device twin)")
        if n:
            with torch.cuda.device(self.device):
                nat.check(nat.lib.ww_wav_batch_decode(self._h, slot, int(bool(normalize)), C.c_void_p(out.data_ptr()),
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    @staticmethod
    def draw_crops(descs, status, n: int = CLIP_SAMPLES) -> None:
        """pad_or_truncate's random crop (:79-81) for every file longer than `n` samples at 16 kHz, python `random` like the reference."""
        n_out = -(-(descs["n_frames"] * descs["up"]) // np.maximum(1, descs["down"]))
        for i in np.nonzero((status == 1) & (n_out > n))[0]:
            descs["crop_start"][i] = random.randint(0, int(n_out[i]) - n)

    def load(self, paths, normalize: bool = True, out: torch.Tensor | None = None, verbose: bool = True, lo: int = 0, hi: int | None = None):
        """read + crop draw + decode of one batch (`paths[lo:hi]`) on the next slot -> (device tensor [B, 16000], ok mask).
        Unreadable files give a zero row and ok False, with the reference's message (:70)."""
        self._not_streaming("load")
        hi = len(paths) if hi is None else hi
        n = hi - lo
        if n > self.max_clips:
            self._regrow(n, max(self.max_raw_bytes, n * 2 * CLIP_SAMPLES + 4096))
        slot = self.next_slot()
        try:
            descs, status = self.read(paths, slot, lo, hi)
        except nat.NativeError as e:
            if e.code != nat.WW_ENOSPACE:
                raise
            self._regrow(max(n, self.max_clips), int(e.needed * 1.25) + 4096)
            slot = self.next_slot()
            descs, status = self.read(paths, slot, lo, hi)
        self.draw_crops(descs, status)
        ok = status == 1
        if verbose and not ok.all():
            names = paths.paths if isinstance(paths, EncodedPaths) else paths
            for i in np.nonzero(~ok)[0][:8]:
                print(f"Error loading {names[lo + i]}: {nat.WAV_STATUS.get(int(status[i]), status[i])}")
        return self.decode(slot, normalize, out), ok

    def _not_streaming(self, what: str) -> None:
        if self._streaming:
            raise RuntimeError(f"WavBatchReader.{what}: a stream() over this reader is still active (one reader serves one consumer at a time: "
                               "its slots and thread pool belong to that stream until it is exhausted or closed) -- use a second reader")

    def stream(self, paths, batch_size: int, normalize: bool = True, verbose: bool = True, start: int = 0):
        """Generator over `paths` in batches of `batch_size`: yields (device tensor [B, 16000], ok mask) per batch.  A helper thread runs
        the reader one batch AHEAD (the ctypes call releases the GIL), the caller's thread draws the crops (python `random`, like every other
        draw of the data path: one thread, so a seeded run repeats -- round 3 drew them on the helper thread, racing the caller's augmentation
        draws), uploads and launches K0 -- the host's file reading then overlaps both the GPU and the caller's own Python work between
        batches.  Needs a reader with >= 3 slots (one being filled, one waiting, one in flight).  A batch that does not fit the staging
        buffer raises NativeError (WW_ENOSPACE, `.needed`): size the reader for the largest batch, or `regrow()` it and call
        stream(..., start=that batch's first index).  While the generator is alive the reader refuses other load() / stream() calls."""
        import queue
        import threading
        if self.slots < 3:
            raise ValueError("stream() needs a reader with at least 3 slots")
        self._not_streaming("stream")
        enc = paths if isinstance(paths, EncodedPaths) else EncodedPaths(paths)
        n = len(enc)
        if batch_size < 1 or batch_size > self.max_clips:
            raise ValueError(f"batch_size {batch_size}: the reader was created for 1..{self.max_clips} files per batch")
        q: queue.Queue = queue.Queue(maxsize=1)
        stop = threading.Event()

        def producer():
            try:
                for lo in range(start, n, batch_size):
                    if stop.is_set():
                        return
                    hi = min(n, lo + batch_size)
                    slot = self.next_slot()
                    descs, status = self.read(enc, slot, lo, hi)
                    q.put((slot, lo, status, descs))
                q.put(None)
            except BaseException as e:              # noqa: BLE001  (handed to the consumer)
                q.put(e)

        t = threading.Thread(target=producer, name="ww-wav-reader", daemon=True)
        self._streaming = True
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                slot, lo, status, descs = item
                self.draw_crops(descs, status)            # the slot's descriptors stay writable until decode() uploads them
                ok = status == 1
                if verbose and not ok.all():
                    for i in np.nonzero(~ok)[0][:8]:
                        print(f"Error loading {enc.paths[lo + i]}: {nat.WAV_STATUS.get(int(status[i]), status[i])}")
                yield (slot if self.host_only else self.decode(slot, normalize)), ok     # host-only readers (tests) hand out the slot number
        finally:
            stop.set()
            while t.is_alive():                      # drain so that a producer blocked on put() can finish
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                t.join(0.001)                        # (a blocking get(timeout) here could sleep its whole timeout after the thread had gone)
            self._streaming = False

    def regrow(self, max_clips: int, max_raw_bytes: int) -> None:
        """Re-create the reader with larger staging (synchronises the device first: nothing may still read the old buffers)."""
        self._regrow(max(max_clips, self.max_clips), max(max_raw_bytes, self.max_raw_bytes))

    def _regrow(self, max_clips: int, max_raw_bytes: int) -> None:
        if not self.host_only:
            torch.cuda.synchronize(self.device)
        self.close()
        self.__init__(max_clips, max_raw_bytes, self.threads, self.slots, self.device, self.host_only)


def probe(path) -> dict | None:
    """Header of one WAV file (no GPU): {'n_frames', 'channels', 'sample_rate', 'format', 'data_offset', 'up', 'down'} or None."""
    d = nat.ClipDesc()
    if nat.lib.ww_wav_probe_host(os.fsencode(path), C.byref(d)) != 1:
        return None
    return {"n_frames": d.n_frames, "channels": d.channels, "sample_rate": d.sample_rate, "format": d.format, "data_offset": d.byte_offset,
            "up": d.up, "down": d.down}
