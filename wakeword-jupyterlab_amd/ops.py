"""Tensor-level entry points: torch custom ops (`torch.ops.wakeword_amd.*`) over the C ABI.

PyTorch is plumbing here (device memory, the current HIP stream, op registration); the arithmetic is
in libwakeword_amd.so.  Every op checks shape / dtype / device, launches on torch's current stream,
never synchronises and never falls back to a CPU implementation.

  logmel(pcm[B,n<=16000] f32, normalize)            -> [B,1,80,32]   (SURVEY.md boundary B1)
  cnn_lstm_forward(x[B,1,80,T<=32], packed, n_conv) -> [B,2]         (boundary B2)
  forward_pcm(pcm[B,n], packed, n_conv, normalize)  -> [B,2]         (boundary B3)
  cnn_pool / lstm_fc                                 the two halves of B2, exposed for tests and profiling
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native as nat
from .config import CLIP_SAMPLES, N_FRAMES, AudioConfig

N_MELS = AudioConfig.N_MELS


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _require_cuda_f32(t: torch.Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if t.device.type != "cuda":
        raise RuntimeError(f"{name} is on {t.device}: this path has no CPU implementation; move it to the MI355X (`.cuda()`)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")


def c_last(n_conv: int) -> int:
    return {2: 64, 3: 128}[n_conv]


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
def pack_state_dict(state_dict) -> np.ndarray:
    """Reference state_dict (torch tensors or numpy arrays, torch layout) -> packed float32 image (host).

    Accepts the key set of SimpleWakewordModel (train_wakeword.py:28-36) or WakewordModel
    (wakeword_training_script.py:141-165); `lstm.weight_hh_l*` may be present and is ignored (it is
    mathematically dead on this path)."""
    def arr(key, shape):
        if key not in state_dict:
            raise KeyError(f"state_dict is missing '{key}'")
        v = state_dict[key]
        v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        v = np.ascontiguousarray(v, dtype=np.float32)
        if tuple(v.shape) != tuple(shape):
            raise ValueError(f"{key}: shape {tuple(v.shape)} != expected {tuple(shape)}")
        return v

    n_conv = 3 if "conv3.weight" in state_dict else 2
    chans = [1, 32, 64, 128][: n_conv + 1]
    keep = []
    sd = nat.StateDict()
    sd.n_conv, sd.hidden = n_conv, 256
    for i in range(n_conv):
        w = arr(f"conv{i + 1}.weight", (chans[i + 1], chans[i], 3, 3)); b = arr(f"conv{i + 1}.bias", (chans[i + 1],))
        keep += [w, b]
        sd.conv_weight[i], sd.conv_bias[i] = w.ctypes.data, b.ctypes.data
    for layer, nin in enumerate([chans[-1], 256]):
        w = arr(f"lstm.weight_ih_l{layer}", (1024, nin))
        bi = arr(f"lstm.bias_ih_l{layer}", (1024,)); bh = arr(f"lstm.bias_hh_l{layer}", (1024,))
        keep += [w, bi, bh]
        sd.lstm_weight_ih[layer], sd.lstm_bias_ih[layer], sd.lstm_bias_hh[layer] = w.ctypes.data, bi.ctypes.data, bh.ctypes.data
    fw = arr("fc.weight", (2, 256)); fb = arr("fc.bias", (2,))
    keep += [fw, fb]
    sd.fc_weight, sd.fc_bias = fw.ctypes.data, fb.ctypes.data
    out = np.empty(nat.check(nat.lib.ww_packed_weights_floats(n_conv)), dtype=np.float32)
    nat.check(nat.lib.ww_pack_weights_host(C.byref(sd), out.ctypes.data))
    return out


def n_conv_of_packed(packed: torch.Tensor) -> int:
    for n_conv in (2, 3):
        if packed.numel() == nat.lib.ww_packed_weights_floats(n_conv):
            return n_conv
    raise ValueError(f"packed weight image of {packed.numel()} floats matches neither model")


# ------------------------------------------------------------------------------------------------
# raw launches (plain functions; the registered custom ops below wrap them)
# ------------------------------------------------------------------------------------------------
def _check_pcm(pcm: torch.Tensor) -> torch.Tensor:
    _require_cuda_f32(pcm, "pcm")
    if pcm.dim() != 2:
        raise ValueError(f"pcm: expected [B, samples], got {tuple(pcm.shape)}")
    if pcm.shape[1] == 0 or pcm.shape[1] > CLIP_SAMPLES:
        raise ValueError(f"pcm: {pcm.shape[1]} samples per clip; the front-end takes 1..{CLIP_SAMPLES} "
                         "(crop longer clips on the host, pad_or_truncate wakeword_training_script.py:78-83)")
    if pcm.stride(1) != 1 or (pcm.shape[0] > 1 and pcm.stride(0) % 4) or pcm.data_ptr() % 16:
        pcm = pcm.contiguous()
        if pcm.shape[1] % 4 and pcm.shape[0] > 1:      # row stride must be a multiple of 4 floats
            pad = torch.zeros(pcm.shape[0], (pcm.shape[1] + 3) // 4 * 4, device=pcm.device, dtype=pcm.dtype)
            pad[:, : pcm.shape[1]] = pcm
            pcm = pad[:, : pcm.shape[1]]
    return pcm


def _logmel_impl(pcm: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    pcm = _check_pcm(pcm)
    B, n = pcm.shape
    out = torch.empty((B, 1, N_MELS, N_FRAMES), device=pcm.device, dtype=torch.float32)
    with torch.cuda.device(pcm.device):
        nat.check(nat.lib.ww_logmel_f32(_ptr(pcm), B, pcm.stride(0) if B > 1 else n, n, int(bool(normalize)), _ptr(out), _stream()))
    return out


def _check_x(x: torch.Tensor) -> torch.Tensor:
    _require_cuda_f32(x, "x")
    if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != N_MELS:
        raise ValueError(f"x: expected [B, 1, {N_MELS}, T], got {tuple(x.shape)}")
    if not 1 <= x.shape[3] <= 32:
        raise NotImplementedError(f"x: T = {x.shape[3]} frames; the conv kernels are built for 1..32 (1 s clips give 32)")
    return x.contiguous()


def _check_packed(packed: torch.Tensor, n_conv: int, like: torch.Tensor) -> None:
    _require_cuda_f32(packed, "packed weights")
    if packed.device != like.device:
        raise RuntimeError(f"packed weights on {packed.device}, input on {like.device}")
    if n_conv not in (2, 3) or packed.numel() != nat.lib.ww_packed_weights_floats(n_conv) or not packed.is_contiguous():
        raise ValueError("packed weights do not match n_conv (use ops.pack_state_dict)")


def _cnn_pool_impl(x: torch.Tensor, packed: torch.Tensor, n_conv: int) -> torch.Tensor:
    x = _check_x(x)
    _check_packed(packed, n_conv, x)
    B, T = x.shape[0], x.shape[3]
    pooled = torch.empty((B, c_last(n_conv)), device=x.device, dtype=torch.float32)
    nbytes = nat.check(nat.lib.ww_cnn_scratch_bytes(B, n_conv))
    scratch = torch.empty(nbytes, device=x.device, dtype=torch.uint8) if nbytes else None
    with torch.cuda.device(x.device):
        nat.check(nat.lib.ww_cnn_pool_f32(_ptr(x), B, T, _ptr(packed), n_conv, _ptr(scratch) if nbytes else None, _ptr(pooled), _stream()))
    return pooled


def _lstm_fc_impl(pooled: torch.Tensor, packed: torch.Tensor, n_conv: int) -> torch.Tensor:
    _require_cuda_f32(pooled, "pooled")
    _check_packed(packed, n_conv, pooled)
    if pooled.dim() != 2 or pooled.shape[1] != c_last(n_conv):
        raise ValueError(f"pooled: expected [B, {c_last(n_conv)}], got {tuple(pooled.shape)}")
    pooled = pooled.contiguous()
    logits = torch.empty((pooled.shape[0], 2), device=pooled.device, dtype=torch.float32)
    with torch.cuda.device(pooled.device):
        nat.check(nat.lib.ww_lstm_fc_f32(_ptr(pooled), pooled.shape[0], _ptr(packed), n_conv, _ptr(logits), _stream()))
    return logits


def _workspace(n: int, n_conv: int, device) -> torch.Tensor:
    return torch.empty(max(1, nat.check(nat.lib.ww_workspace_bytes(n, n_conv))), device=device, dtype=torch.uint8)


def _cnn_lstm_forward_impl(x: torch.Tensor, packed: torch.Tensor, n_conv: int) -> torch.Tensor:
    x = _check_x(x)
    _check_packed(packed, n_conv, x)
    B, T = x.shape[0], x.shape[3]
    logits = torch.empty((B, 2), device=x.device, dtype=torch.float32)
    ws = _workspace(B, n_conv, x.device)
    with torch.cuda.device(x.device):
        nat.check(nat.lib.ww_model_forward_f32(_ptr(x), B, T, _ptr(packed), n_conv, _ptr(ws), _ptr(logits), _stream()))
    return logits


def _forward_pcm_impl(pcm: torch.Tensor, packed: torch.Tensor, n_conv: int, normalize: bool = True) -> torch.Tensor:
    pcm = _check_pcm(pcm)
    _check_packed(packed, n_conv, pcm)
    B, n = pcm.shape
    logits = torch.empty((B, 2), device=pcm.device, dtype=torch.float32)
    ws = _workspace(B, n_conv, pcm.device)
    with torch.cuda.device(pcm.device):
        nat.check(nat.lib.ww_forward_pcm_f32(_ptr(pcm), B, pcm.stride(0) if B > 1 else n, n, int(bool(normalize)),
                                             _ptr(packed), n_conv, _ptr(ws), _ptr(logits), _stream()))
    return logits


# ------------------------------------------------------------------------------------------------
# torch custom ops: torch.ops.wakeword_amd.{logmel,cnn_pool,lstm_fc,cnn_lstm_forward,forward_pcm}
# ------------------------------------------------------------------------------------------------
# COMPILED operators (csrc/ww_torch_ops.cpp -> libwakeword_amd_torch.so: TORCH_LIBRARY schema + CUDA, Meta and CPU kernels, SURVEY.md
# section 7 step 2).  The CUDA kernels validate, allocate with ATen and call the C ABI on torch's current stream; the Meta kernels give
# shapes only (FakeTensor / torch.compile tracing of the drop-in modules); the CPU kernels refuse.  Rounds 1-3 registered the Python
# functions above through torch.library; those stay as plain functions (`_logmel_impl` ...: tests and the raw launches of bench.py's legs).
# The library does not link libwakeword_amd.so: it is handed the addresses of the nine C ABI functions it calls.
# No fallback: without the compiled library the package does not import.
import os as _os_ops

TORCH_LIB_PATH = _os_ops.path.join(_os_ops.path.dirname(_os_ops.path.abspath(__file__)), "libwakeword_amd_torch.so")
if not _os_ops.path.exists(TORCH_LIB_PATH):
    raise ImportError(f"{TORCH_LIB_PATH} is missing: build it with `make -C wakeword-jupyterlab_amd/csrc` "
                      "(or `python -c 'import __graft_entry__ as g; g.build()'`)")
torch.ops.load_library(TORCH_LIB_PATH)
# bind the operators to THE copy of libwakeword_amd.so this process uses (the shipped one or a WW_LIB_OVERRIDE build): addresses, not names
_TORCH_BIND_ORDER = ("ww_last_error", "ww_packed_weights_floats", "ww_cnn_scratch_bytes", "ww_workspace_bytes", "ww_logmel_f32",
                     "ww_cnn_pool_f32", "ww_lstm_fc_f32", "ww_model_forward_f32", "ww_forward_pcm_f32")
_torch_lib = C.CDLL(TORCH_LIB_PATH)
_table = (C.c_void_p * len(_TORCH_BIND_ORDER))(*[C.cast(getattr(nat.lib, _n), C.c_void_p).value for _n in _TORCH_BIND_ORDER])
if _torch_lib.ww_torch_bind(_table, len(_TORCH_BIND_ORDER)) != 0:
    raise ImportError("libwakeword_amd_torch.so refused the C ABI table (rebuild: make -C wakeword-jupyterlab_amd/csrc)")


# The public functions check argument types in Python first (TypeError / ValueError / NotImplementedError with the messages rounds 1-3
# gave; the compiled kernels re-check with TORCH_CHECK -> RuntimeError), then dispatch: real tensors reach the HIP kernels, fake / meta
# tensors the shape-only kernels.
def _precheck_pcm(pcm) -> None:
    _require_cuda_f32(pcm, "pcm")
    if pcm.dim() != 2:
        raise ValueError(f"pcm: expected [B, samples], got {tuple(pcm.shape)}")
    if pcm.shape[1] == 0 or pcm.shape[1] > CLIP_SAMPLES:
        raise ValueError(f"pcm: {pcm.shape[1]} samples per clip; the front-end takes 1..{CLIP_SAMPLES} "
                         "(crop longer clips on the host, pad_or_truncate wakeword_training_script.py:78-83)")


def _precheck_x(x) -> None:
    _require_cuda_f32(x, "x")
    if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != N_MELS:
        raise ValueError(f"x: expected [B, 1, {N_MELS}, T], got {tuple(x.shape)}")
    if not 1 <= x.shape[3] <= 32:
        raise NotImplementedError(f"x: T = {x.shape[3]} frames; the conv kernels are built for 1..32 (1 s clips give 32)")


def logmel(pcm: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    _precheck_pcm(pcm)
    return torch.ops.wakeword_amd.logmel(pcm, bool(normalize))


def cnn_pool(x, packed, n_conv):
    _precheck_x(x)
    _check_packed(packed, n_conv, x)
    return torch.ops.wakeword_amd.cnn_pool(x, packed, n_conv)


def lstm_fc(pooled, packed, n_conv):
    _require_cuda_f32(pooled, "pooled")
    _check_packed(packed, n_conv, pooled)
    if pooled.dim() != 2 or pooled.shape[1] != c_last(n_conv):
        raise ValueError(f"pooled: expected [B, {c_last(n_conv)}], got {tuple(pooled.shape)}")
    return torch.ops.wakeword_amd.lstm_fc(pooled, packed, n_conv)


def cnn_lstm_forward(x, packed, n_conv):
    _precheck_x(x)
    _check_packed(packed, n_conv, x)
    return torch.ops.wakeword_amd.cnn_lstm_forward(x, packed, n_conv)


def forward_pcm(pcm, packed, n_conv, normalize: bool = True):
    _precheck_pcm(pcm)
    _check_packed(packed, n_conv, pcm)
    return torch.ops.wakeword_amd.forward_pcm(pcm, packed, n_conv, bool(normalize))


# ------------------------------------------------------------------------------------------------
# training step (SimpleWakewordModel): train-mode forward + backward on the HIP kernels of csrc/ww_train.hip
# ------------------------------------------------------------------------------------------------
def _train_keys(n_conv):
    convs = [k for i in range(1, n_conv + 1) for k in (f"conv{i}.weight", f"conv{i}.bias")]
    return tuple(convs + ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
                          "lstm.weight_ih_l1", "lstm.weight_hh_l1", "lstm.bias_ih_l1", "lstm.bias_hh_l1", "fc.weight", "fc.bias"])


def _train_params_struct(params, n_conv):
    convs, (wi0, _wh0, bi0, bh0, wi1, _wh1, bi1, bh1, fw, fb) = params[:2 * n_conv], params[2 * n_conv:]
    tp = nat.TrainParams()
    tp.n_conv, tp.hidden = n_conv, 256
    for i in range(n_conv):
        tp.conv_weight[i], tp.conv_bias[i] = convs[2 * i].data_ptr(), convs[2 * i + 1].data_ptr()
    tp.lstm_weight_ih[0], tp.lstm_weight_ih[1] = wi0.data_ptr(), wi1.data_ptr()
    tp.lstm_bias_ih[0], tp.lstm_bias_ih[1] = bi0.data_ptr(), bi1.data_ptr()
    tp.lstm_bias_hh[0], tp.lstm_bias_hh[1] = bh0.data_ptr(), bh1.data_ptr()
    tp.fc_weight, tp.fc_bias = fw.data_ptr(), fb.data_ptr()
    return tp


class _TrainStep(torch.autograd.Function):
    """logits = model(x) in train mode, with d loss / d parameters from the HIP backward kernels.  `x` gets no gradient (the
    reference never asks for one)."""
    # diagnostics only (train_last_masks / _packed_image / _bit_images): a WEAK reference to the newest step's workspace, so that it
    # does not outlive its step (a strong one kept 2.7 GB alive behind every fp32 step); tests that read it after the backward
    # switch keep_train_workspace(True) on
    _last_ws_ref = None
    _last_ws_keep = None
    keep = False
    last_n_conv = 2
    last_n = 0

    @staticmethod
    def forward(ctx, x, n_conv, p_lstm, p_fc, seed, *params):
        x = _check_x(x)
        params = tuple(p.detach().contiguous() for p in params)
        for p in params:
            _require_cuda_f32(p, "parameter")
        B, T = x.shape[0], x.shape[3]
        if B == 0:
            raise ValueError("empty training batch")
        logits = torch.empty((B, 2), device=x.device, dtype=torch.float32)
        math = nat.lib.ww_get_train_math()                  # resolved ONCE: query, forward and backward of this step all get this value
        with torch.cuda.device(x.device):
            ws = torch.empty(nat.check(nat.lib.ww_train_workspace_bytes(B, n_conv, math)), device=x.device, dtype=torch.uint8)
            tp = _train_params_struct(params, n_conv)
            nat.check(nat.lib.ww_train_forward_f32(_ptr(x), B, T, C.byref(tp), float(p_lstm), float(p_fc), int(seed) & (2 ** 64 - 1),
                                                   math, _ptr(ws), ws.numel(), _ptr(logits), _stream()))
        ctx.save_for_backward(x, ws, *params)
        ctx.n_conv = n_conv
        ctx.train_math = math
        import weakref
        _TrainStep._last_ws_ref, _TrainStep.last_n_conv, _TrainStep.last_n = weakref.ref(ws), n_conv, B
        _TrainStep._last_ws_keep = ws if _TrainStep.keep else None
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        x, ws, *params = ctx.saved_tensors
        n_conv = ctx.n_conv
        B, T = x.shape[0], x.shape[3]
        dlogits = dlogits.contiguous().float()
        grads = [torch.empty_like(p) for p in params]
        o = 2 * n_conv                      # index of lstm.weight_ih_l0
        grads[o + 1].zero_()                # weight_hh: h0 = 0, the gradient is exactly zero
        grads[o + 5].zero_()
        tg = nat.TrainGrads()
        for i in range(n_conv):
            tg.conv_weight[i], tg.conv_bias[i] = grads[2 * i].data_ptr(), grads[2 * i + 1].data_ptr()
        tg.lstm_weight_ih[0], tg.lstm_bias[0] = grads[o].data_ptr(), grads[o + 2].data_ptr()
        tg.lstm_weight_ih[1], tg.lstm_bias[1] = grads[o + 4].data_ptr(), grads[o + 6].data_ptr()
        tg.fc_weight, tg.fc_bias = grads[o + 8].data_ptr(), grads[o + 9].data_ptr()
        with torch.cuda.device(x.device):
            tp = _train_params_struct(params, n_conv)
            nat.check(nat.lib.ww_train_backward_f32(_ptr(x), B, T, C.byref(tp), _ptr(dlogits), ctx.train_math, _ptr(ws), ws.numel(),
                                                    C.byref(tg), _stream()))
        grads[o + 3].copy_(grads[o + 2])    # d/d bias_hh == d/d bias_ih
        grads[o + 7].copy_(grads[o + 6])
        return (None, None, None, None, None, *grads)


def train_forward(x, named_params: dict, n_conv: int, p_lstm: float, p_fc: float, seed: int):
    """x [B,1,80,T] + the module's parameters (reference key set) -> logits [B,2] with an autograd graph behind them."""
    return _TrainStep.apply(x, n_conv, p_lstm, p_fc, seed, *[named_params[k] for k in _train_keys(n_conv)])


def keep_train_workspace(on: bool = True) -> None:
    """Diagnostics switch: hold a strong reference to the newest training step's workspace so that train_last_* can read it after
    the step's autograd graph is gone (off by default: the workspace then dies with its step)."""
    _TrainStep.keep = bool(on)
    if not on:
        _TrainStep._last_ws_keep = None


def _last_workspace() -> torch.Tensor:
    ws = _TrainStep._last_ws_ref() if _TrainStep._last_ws_ref is not None else None
    if ws is None:
        raise RuntimeError("no live training workspace: the last step's graph was freed (ops.keep_train_workspace(True) keeps it for diagnostics)")
    return ws


def train_last_masks(n: int):
    """Dropout factors of the most recent training forward: (mask0 [n,256], mask1 [n,256]) -- tests replay them elsewhere."""
    ws = _last_workspace()
    m0 = torch.empty((n, 256), device=ws.device, dtype=torch.float32)
    m1 = torch.empty_like(m0)
    with torch.cuda.device(ws.device):
        nat.check(nat.lib.ww_train_masks(_ptr(ws), n, _TrainStep.last_n_conv, _ptr(m0), _ptr(m1), _stream()))
    return m0, m1


def train_last_packed_image() -> torch.Tensor:
    """Diagnostic: the packed image the most recent split-precision training forward of the 2-conv model wrote on the device."""
    ws = _last_workspace()
    img = torch.empty(int(nat.lib.ww_packed_weights_floats(_TrainStep.last_n_conv)), device=ws.device, dtype=torch.float32)
    with torch.cuda.device(ws.device):
        nat.check(nat.lib.ww_train_packed_image(_ptr(ws), _TrainStep.last_n, _TrainStep.last_n_conv, _ptr(img), _stream()))
    return img


def train_last_bit_images():
    """Diagnostic: (mask_last uint8 [n,80,32,C/8], sign1 int32 [n,80,32]) of the most recent split-precision training forward."""
    ws, n, nc = _last_workspace(), _TrainStep.last_n, _TrainStep.last_n_conv
    mask = torch.empty((n, 80, 32, 8 if nc == 2 else 16), device=ws.device, dtype=torch.uint8)
    sign1 = torch.empty((n, 80, 32), device=ws.device, dtype=torch.int32)
    with torch.cuda.device(ws.device):
        nat.check(nat.lib.ww_train_bit_images(_ptr(ws), n, nc, _ptr(mask), _ptr(sign1), _stream()))
    return mask, sign1


CONV_MATH = {"f32": 0, "f16x3": 1, "f16x3d": 2}


def set_conv_math(mode: str) -> None:
    """Arithmetic of the conv / LSTM-gate GEMMs, process-wide: 'f32' (exact fp32 MFMA), 'f16x3' (each fp32 operand as two f16
    halves, three f16 MFMAs per product block, fp32 accumulate; ~2^-21 relative error, 3/16 the MFMA cycles; the 2-conv model's
    conv2 as a 1-D Winograd F(2,3)) or 'f16x3d' (f16x3 with every convolution in its direct form)."""
    if mode not in CONV_MATH:
        raise ValueError(f"conv math {mode!r}: expected one of {sorted(CONV_MATH)}")
    nat.check(nat.lib.ww_set_conv_math(CONV_MATH[mode]))


def set_conv_math_thread(mode) -> None:
    """Override the conv arithmetic for launches issued by the CALLING THREAD only (None clears it): a streamer thread and a batch
    job can then use different arithmetics safely."""
    if mode is not None and mode not in CONV_MATH:
        raise ValueError(f"conv math {mode!r}: expected one of {sorted(CONV_MATH)} or None")
    nat.check(nat.lib.ww_set_conv_math_thread(-1 if mode is None else CONV_MATH[mode]))


def get_conv_math() -> str:
    return {v: k for k, v in CONV_MATH.items()}[nat.lib.ww_get_conv_math()]


LOGMEL_MATH = {"f32": 0, "f64": 1, "auto": 2}


def set_logmel_math(mode: str) -> None:
    """Arithmetic of the log-mel front end, process-wide: 'f32' (float32 FFT), 'f64' (float64 window product / FFT / split, what
    the reference's numpy.fft.rfft computes) or 'auto' (default: float32, and the clips with a live mel band on the float32
    FFT's rounding floor are redone in float64)."""
    if mode not in LOGMEL_MATH:
        raise ValueError(f"log-mel math {mode!r}: expected one of {sorted(LOGMEL_MATH)}")
    nat.check(nat.lib.ww_set_logmel_math(LOGMEL_MATH[mode]))


def set_logmel_math_thread(mode) -> None:
    """Per-thread override of the log-mel arithmetic (None clears it); see set_conv_math_thread."""
    if mode is not None and mode not in LOGMEL_MATH:
        raise ValueError(f"log-mel math {mode!r}: expected one of {sorted(LOGMEL_MATH)} or None")
    nat.check(nat.lib.ww_set_logmel_math_thread(-1 if mode is None else LOGMEL_MATH[mode]))


def get_logmel_math() -> str:
    return {v: k for k, v in LOGMEL_MATH.items()}[nat.lib.ww_get_logmel_math()]


TRAIN_MATH = {"f32": 0, "f16x3": 1}


def set_train_math(mode: str) -> None:
    """Arithmetic of the training step's conv kernels, process-wide: 'f16x3' (default: split precision on the f16 matrix
    instructions where a kernel exists -- SimpleWakewordModel's conv2 backward) or 'f32' (exact fp32 matrix instructions)."""
    if mode not in TRAIN_MATH:
        raise ValueError(f"train math {mode!r}: expected one of {sorted(TRAIN_MATH)}")
    nat.check(nat.lib.ww_set_train_math(TRAIN_MATH[mode]))


def get_train_math() -> str:
    return {v: k for k, v in TRAIN_MATH.items()}[nat.lib.ww_get_train_math()]


def init() -> None:
    """Upload the front-end tables for the current device (needed before hipGraph capture)."""
    nat.check(nat.lib.ww_init())


import os as _os  # noqa: E402

if _os.environ.get("WW_CONV_MATH"):
    set_conv_math(_os.environ["WW_CONV_MATH"])
if _os.environ.get("WW_LOGMEL_MATH"):
    set_logmel_math(_os.environ["WW_LOGMEL_MATH"])


# ---- KA: augmentation ------------------------------------------------------------------
def augment(pcm: torch.Tensor, plans) -> torch.Tensor:
    """pcm [B, 16000] float32 on the GPU + one plan per clip -> augmented [B, 16000] (ww_augment_f32).

    `plans`: a ctypes array of _native.AugmentPlan, or a list of dicts with the keys of oracle-style plans
    (shift, n_steps | pitch_rate, rate, crop, sigma, seed); see AudioProcessor.draw_augment_plan."""
    import ctypes as C
    if pcm.device.type != "cuda":
        raise RuntimeError("augment: pcm must live on the MI355X (no CPU fallback)")
    if pcm.dtype != torch.float32 or pcm.dim() != 2 or pcm.shape[1] != 16000:
        raise ValueError(f"augment: expected float32 [B, 16000], got {pcm.dtype} {tuple(pcm.shape)}")
    pcm = pcm.contiguous()
    B = pcm.shape[0]
    if not isinstance(plans, C.Array):
        arr = (nat.AugmentPlan * max(1, B))()
        if len(plans) != B:
            raise ValueError(f"augment: {len(plans)} plans for {B} clips")
        for i, p in enumerate(plans):
            a = arr[i]
            a.shift = int(p.get("shift", 0))
            a.crop_start = int(p.get("crop", 0))
            n_steps = p.get("n_steps")
            a.pitch_rate = float(p["pitch_rate"]) if p.get("pitch_rate") else (2.0 ** (-float(n_steps) / 12.0) if n_steps is not None else 0.0)
            a.stretch_rate = float(p["rate"]) if p.get("rate") else 0.0
            a.noise_sigma = float(p.get("sigma", 0.0))
            a.noise_seed = int(p.get("seed", 0)) & 0xFFFFFFFF
        plans = arr
    elif len(plans) < B:
        raise ValueError(f"augment: {len(plans)} plans for {B} clips")
    out = torch.empty_like(pcm)
    if B == 0:
        return out
    with torch.cuda.device(pcm.device):
        ws_bytes = nat.check(nat.lib.ww_augment_workspace_bytes(B))
        ws = torch.empty(ws_bytes, device=pcm.device, dtype=torch.uint8)
        nat.check(nat.lib.ww_augment_f32(C.c_void_p(pcm.data_ptr()), B, pcm.stride(0), plans, C.c_void_p(out.data_ptr()),
                                         C.c_void_p(ws.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        ws.record_stream(torch.cuda.current_stream())
    return out
