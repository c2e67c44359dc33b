"""wakeword-jupyterlab_amd: MI355X-native log-mel + CNN+LSTM wakeword inference path.

Drop-in for ONE hot path of sarpel/wakeword-jupyterlab -- 16 kHz PCM -> STFT -> 80-mel -> log-mel dB ->
Conv2d+ReLU stack -> global average pool -> 2-layer LSTM step -> Linear -> logits -- as hand-written
HIP kernels for gfx950 behind the reference's own Python signatures.  See DESIGN.md.

Submodules that need libwakeword_amd.so (ops, model, audio, dataset, streaming, inference) fail loudly at
import when it has not been built; `synth`, `config` and `distributed` are pure host code.
"""
__version__ = "0.1.0"

from . import config, synth  # noqa: F401  (no native dependency)


def __getattr__(name):
    # lazy: `import wakeword_jupyterlab_amd` must work on a box where only host utilities are needed
    import importlib
    if name in ("ops", "model", "audio", "dataset", "streaming", "inference", "distributed", "_native"):
        return importlib.import_module(f"{__name__}.{name}")
    lazy = {"AudioProcessor": "audio", "WakewordDataset": "dataset", "DataLoader": "dataset", "SimpleWakewordModel": "model",
            "WakewordModel": "model", "StreamingDetector": "streaming", "predict_wakeword": "inference",
            "evaluate": "inference", "AudioConfig": "config", "ModelConfig": "config", "Config": "config",
            "AugmentationConfig": "config"}
    if name in lazy:
        return getattr(importlib.import_module(f"{__name__}.{lazy[name]}"), name)
    raise AttributeError(name)
