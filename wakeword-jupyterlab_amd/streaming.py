"""Streaming sliding-window detection: many microphones, one hop per step, hipGraph-replayed.

The reference has no streaming code; the per-window semantics are `predict_wakeword` (notebook cell 19,
wakeword_training.ipynb:871-893): peak-normalise the last 1 s, log-mel, forward, softmax, p[1] >= 0.8.
Per hop this keeps a 16000-sample ring per microphone in HBM, appends the hop, recomputes the whole window
(frames cannot be reused across 10 ms hops: the 512-sample STFT grid realigns only every 2560 samples and
`ref=np.max` is per window -- SURVEY.md section 7) and replays one captured hipGraph:
ring append -> K1 -> K2 -> K3 (+softmax).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as nat
from .config import CLIP_SAMPLES


class StreamingDetector:
    def __init__(self, model, n_mics: int = 256, hop_samples: int = 160, threshold: float = 0.8, device=None):
        if model.training:
            raise NotImplementedError("call model.eval() first")
        self.device = torch.device(device) if device is not None else model.fc.weight.device
        if self.device.type != "cuda":
            raise RuntimeError("StreamingDetector needs the model on the MI355X (no CPU path)")
        self.n_mics, self.hop, self.threshold = int(n_mics), int(hop_samples), float(threshold)
        self._packed = model.packed_weights()          # keep alive: the graph holds its pointer
        self._n_conv = model._n_conv
        self._stream = torch.cuda.Stream(device=self.device)
        self.hop_buf = torch.zeros((self.n_mics, self.hop), device=self.device, dtype=torch.float32)
        self.prob = torch.zeros(self.n_mics, device=self.device, dtype=torch.float32)
        self.logits = torch.zeros((self.n_mics, 2), device=self.device, dtype=torch.float32)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(nat.lib.ww_streamer_create(self.n_mics, self.hop, C.c_void_p(self._packed.data_ptr()), self._n_conv,
                                                 C.c_void_p(self._stream.cuda_stream), C.byref(handle)))
        self._h = handle

    @property
    def stream(self) -> torch.cuda.Stream:
        return self._stream

    def step(self, hop: torch.Tensor | None = None) -> torch.Tensor:
        """Push one hop [n_mics, hop_samples] (or reuse whatever is in `hop_buf`) and enqueue the graph.
        Returns `self.prob` (softmax p(wakeword) per mic), valid once `self.stream` has caught up."""
        if hop is not None:
            if hop.device.type == "cuda":
                # `hop` was produced on the caller's stream: order our stream behind it before reading it
                self._stream.wait_stream(torch.cuda.current_stream(self.device))
                hop.record_stream(self._stream)
            with torch.cuda.stream(self._stream):
                self.hop_buf.copy_(hop, non_blocking=True)
        with torch.cuda.device(self.device):
            nat.check(nat.lib.ww_streamer_step(self._h, C.c_void_p(self.hop_buf.data_ptr()), C.c_void_p(self.prob.data_ptr()),
                                               C.c_void_p(self.logits.data_ptr())))
        return self.prob

    def detections(self) -> torch.Tensor:
        self._stream.synchronize()
        return self.prob >= self.threshold

    def window(self) -> torch.Tensor:
        """Current 1 s window of every microphone, oldest sample first: [n_mics, 16000]."""
        out = torch.empty((self.n_mics, CLIP_SAMPLES), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            nat.check(nat.lib.ww_streamer_window(self._h, C.c_void_p(out.data_ptr())))
        self._stream.synchronize()
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._stream.synchronize()
            nat.lib.ww_streamer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
