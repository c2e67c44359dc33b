"""ctypes binding of libwakeword_amd.so (the C ABI declared in include/wakeword_amd.h).

The shared library is the ONLY implementation of the hot path: importing this module fails loudly
when it has not been built (`python -c "import __graft_entry__ as g; g.build()"` or
`make -C wakeword-jupyterlab_amd/csrc`), and every launch fails with WW_ENODEVICE when no gfx950
device is visible.  There is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

# torch first: it carries its own libamdhip64 (SONAME libamdhip64.so.7).  Loading ours afterwards makes the
# dynamic loader bind libwakeword_amd.so to that SAME runtime instance, so torch's streams, allocations and
# device pointers are valid inside the library.  The other order would leave two HIP runtimes in the process.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WW_LIB_OVERRIDE") or os.path.join(_HERE, "libwakeword_amd.so")   # override: ablation builds only

WW_OK, WW_EINVAL, WW_ENODEVICE, WW_EHIP, WW_EUNSUPPORTED, WW_ENOSPACE = 0, -1, -2, -3, -4, -5
ABI_VERSION = 4


class NativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libwakeword_amd: {msg} (code {code})")
        self.code = code


class StateDict(C.Structure):
    """struct ww_state_dict (include/wakeword_amd.h)."""
    _fields_ = [
        ("n_conv", C.c_int32), ("hidden", C.c_int32),
        ("conv_weight", C.c_void_p * 3), ("conv_bias", C.c_void_p * 3),
        ("lstm_weight_ih", C.c_void_p * 2), ("lstm_bias_ih", C.c_void_p * 2), ("lstm_bias_hh", C.c_void_p * 2),
        ("fc_weight", C.c_void_p), ("fc_bias", C.c_void_p),
    ]


class TrainParams(C.Structure):
    """struct ww_train_params: DEVICE pointers, torch layout."""
    _fields_ = [
        ("n_conv", C.c_int32), ("hidden", C.c_int32),
        ("conv_weight", C.c_void_p * 3), ("conv_bias", C.c_void_p * 3),
        ("lstm_weight_ih", C.c_void_p * 2), ("lstm_bias_ih", C.c_void_p * 2), ("lstm_bias_hh", C.c_void_p * 2),
        ("fc_weight", C.c_void_p), ("fc_bias", C.c_void_p),
    ]


class TrainGrads(C.Structure):
    """struct ww_train_grads: DEVICE output pointers."""
    _fields_ = [
        ("conv_weight", C.c_void_p * 3), ("conv_bias", C.c_void_p * 3),
        ("lstm_weight_ih", C.c_void_p * 2), ("lstm_bias", C.c_void_p * 2),
        ("fc_weight", C.c_void_p), ("fc_bias", C.c_void_p),
    ]


class ClipDesc(C.Structure):
    """struct ww_clip_desc (include/wakeword_amd.h)."""
    _fields_ = [
        ("byte_offset", C.c_int64), ("n_frames", C.c_int64), ("channels", C.c_int32), ("sample_rate", C.c_int32),
        ("format", C.c_int32), ("crop_start", C.c_int32), ("up", C.c_int32), ("down", C.c_int32),
        ("half_len", C.c_int32), ("_pad", C.c_int32), ("taps_dev", C.c_void_p),
    ]


class AugmentPlan(C.Structure):
    """struct ww_augment_plan (include/wakeword_amd.h)."""
    _fields_ = [
        ("shift", C.c_int32), ("crop_start", C.c_int32), ("pitch_rate", C.c_double), ("stretch_rate", C.c_double),
        ("noise_sigma", C.c_float), ("noise_seed", C.c_uint32),
    ]


FMT_S16, FMT_S24, FMT_S32, FMT_F32, FMT_U8, FMT_F64 = 1, 2, 3, 4, 5, 6
WAV_STATUS = {1: "ok", -1: "cannot open", -2: "not a RIFF/WAVE file", -3: "missing fmt/data chunk", -4: "unsupported WAV encoding",
              -5: "read error", -6: "staging buffer full"}

# name -> (restype, argtypes); kept in one table so tests can check it against the header
PROTOTYPES = {
    "ww_abi_version": (C.c_int, []),
    "ww_last_error": (C.c_char_p, []),
    "ww_init": (C.c_int, []),
    "ww_set_conv_math": (C.c_int, [C.c_int]),
    "ww_get_conv_math": (C.c_int, []),
    "ww_set_conv_math_thread": (C.c_int, [C.c_int]),
    "ww_set_logmel_math_thread": (C.c_int, [C.c_int]),
    "ww_set_train_math": (C.c_int, [C.c_int]),
    "ww_get_train_math": (C.c_int, []),
    "ww_set_logmel_math": (C.c_int, [C.c_int]),
    "ww_get_logmel_math": (C.c_int, []),
    "ww_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "ww_sync_timeouts": (C.c_int, []),
    "ww_mel_filterbank_host": (C.c_int, [C.c_void_p]),
    "ww_hann_window_host": (C.c_int, [C.c_void_p]),
    "ww_resample_taps_host": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ww_resampler_prepare": (C.c_int, [C.c_int32, C.POINTER(ClipDesc)]),
    "ww_decode_resample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "ww_wav_reader_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    "ww_wav_reader_staging": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "ww_wav_probe_host": (C.c_int, [C.c_char_p, C.POINTER(ClipDesc)]),
    "ww_wav_reader_destroy": (C.c_int, [C.c_void_p]),
    "ww_read_wav_batch_host": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int64, C.c_int32, C.POINTER(C.POINTER(ClipDesc)), C.c_void_p,
                                         C.POINTER(C.c_int64)]),
    "ww_wav_batch_decode": (C.c_int, [C.c_void_p, C.c_int32, C.c_int, C.c_void_p, C.c_void_p]),
    "ww_augment_workspace_bytes": (C.c_int64, [C.c_int64]),
    "ww_augment_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(AugmentPlan), C.c_void_p, C.c_void_p, C.c_void_p]),
    "ww_augment_record_bytes": (C.c_int64, []),
    "ww_augment_plans_prepare": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "ww_augment_records_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ww_kaiser_best_host": (C.c_int, [C.c_void_p]),
    "ww_logmel_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "ww_packed_weights_floats": (C.c_int64, [C.c_int32]),
    "ww_pack_weights_host": (C.c_int, [C.POINTER(StateDict), C.c_void_p]),
    "ww_cnn_scratch_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "ww_cnn_pool_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ww_lstm_fc_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ww_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "ww_model_forward_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ww_forward_pcm_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ww_train_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "ww_train_forward_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(TrainParams), C.c_float, C.c_float, C.c_uint64, C.c_int32,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ww_train_backward_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(TrainParams), C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                        C.POINTER(TrainGrads), C.c_void_p]),
    "ww_train_masks": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ww_train_packed_image": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "ww_train_bit_images": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ww_streamer_create": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ww_streamer_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ww_streamer_window": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ww_streamer_destroy": (C.c_int, [C.c_void_p]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library is the only implementation of this path (no CPU "
            "fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C wakeword-jupyterlab_amd/csrc`.")
    # RTLD_LOCAL (the default): nothing of this library enters the global symbol scope.  HIP kernel stubs have default visibility, so a
    # globally loaded copy would interpose the kernels of every other build opened later in the process (scripts/ab_kernels.py opens
    # several); libwakeword_amd_torch.so gets the addresses it needs from this handle instead (ops.py -> ww_torch_bind).
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype, fn.argtypes = res, args
    if lib.ww_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI {lib.ww_abi_version()} != expected {ABI_VERSION}; rebuild")
    return lib


lib = _load()


def check(rc: int) -> int:
    """Raise NativeError on a negative return code; pass sizes / WW_OK through."""
    if rc < 0:
        raise NativeError(rc, (lib.ww_last_error() or b"").decode("utf-8", "replace"))
    return rc


def device_info():
    n_cu, khz = C.c_int(0), C.c_int(0)
    name = C.create_string_buffer(128)
    check(lib.ww_device_info(C.byref(n_cu), C.byref(khz), name, 128))
    return {"n_cu": n_cu.value, "clock_khz": khz.value, "name": name.value.decode()}
