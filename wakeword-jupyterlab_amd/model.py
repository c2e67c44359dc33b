"""Drop-in nn.Modules for the reference models, with `forward` on the HIP kernels.

  SimpleWakewordModel  <- /root/reference/wakeword_training/train_wakeword.py:28-49   (2 convs; graded model)
  WakewordModel        <- /root/reference/wakeword_training_script.py:141-184          (3 convs; notebook cell 7)

Same constructor signatures, same parameter names and shapes (so `state_dict()` /
`load_state_dict(ckpt['model_state_dict'])` interchange with the reference, `lstm.weight_hh_l*`
included), same `forward(x[B,1,80,T]) -> logits[B,2]`.  The parameters live in ordinary torch layers;
`forward` packs them once per weight version into the kernel layout (MFMA B-operand order, transposed
W_ih without the dead forget gate, pre-summed biases) and calls `torch.ops.wakeword_amd.cnn_lstm_forward`.

Training: in train mode both modules run the train-mode forward (inter-layer LSTM dropout, dropout before
fc) and, through a `torch.autograd.Function`, the HIP backward kernels, so the reference's loops
(`output = model(data); loss = criterion(output, target); loss.backward(); optimizer.step()`, train_wakeword.py:109-115,
WakewordTrainer.train_epoch wakeword_training_script.py:241-267) work unchanged.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .config import AudioConfig, Config, ModelConfig


class _CnnLstm(nn.Module):
    def __init__(self, channels, hidden, num_layers, dropout, num_classes):
        super().__init__()
        if hidden != 256 or num_layers != 2 or num_classes != 2:
            raise NotImplementedError("the HIP head is built for hidden 256, 2 LSTM layers, 2 classes (the reference configs)")
        for i in range(1, len(channels)):
            setattr(self, f"conv{i}", nn.Conv2d(channels[i - 1], channels[i], kernel_size=3, padding=1))
        self.pool = nn.AdaptiveAvgPool2d((1, 1))
        self.lstm = nn.LSTM(input_size=channels[-1], hidden_size=hidden, num_layers=num_layers, batch_first=True,
                            dropout=dropout if num_layers > 1 else 0)
        self.dropout = nn.Dropout(dropout)
        self.fc = nn.Linear(hidden, num_classes)
        self._n_conv = len(channels) - 1
        self._packed = None
        self._packed_key = None

    # ---- packed weights: rebuilt whenever a parameter changes in place, moves, or is reloaded ----
    def _weights_key(self):
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in self.parameters())

    def packed_weights(self) -> torch.Tensor:
        key = self._weights_key()
        if self._packed is None or key != self._packed_key:
            dev = self.fc.weight.device
            if dev.type != "cuda":
                raise RuntimeError("model parameters are on the CPU: this path has no CPU implementation; call .to('cuda')")
            host = ops.pack_state_dict(self.state_dict())
            self._packed = torch.from_numpy(host).to(dev)
            self._packed_key = key
        return self._packed

    def forward(self, x):
        if self.training:
            if self.fc.weight.device.type != "cuda":
                raise RuntimeError("model parameters are on the CPU: this path has no CPU implementation; call .to('cuda')")
            # dropout factors come from a counter-based generator; its seed is drawn from torch's CPU generator, so
            # torch.manual_seed() makes a run repeatable
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            return ops.train_forward(x, dict(self.named_parameters()), self._n_conv, float(self.lstm.dropout), float(self.dropout.p), seed)
        return ops.cnn_lstm_forward(x, self.packed_weights(), self._n_conv)

    def forward_pcm(self, pcm, normalize: bool = True):
        """PCM [B, n<=16000] -> logits [B, 2]: the Dataset's mel path and forward() in one call (boundary B3)."""
        if self.training:
            raise NotImplementedError("call model.eval() first")
        return ops.forward_pcm(pcm, self.packed_weights(), self._n_conv, normalize)


class SimpleWakewordModel(_CnnLstm):
    """train_wakeword.py:28-36: Conv(1,32) Conv(32,64) pool LSTM(64,256,2,dropout .5) Linear(256,2)."""

    def __init__(self):
        super().__init__([1, 32, 64], Config.HIDDEN_SIZE, Config.NUM_LAYERS, Config.DROPOUT, 2)


class WakewordModel(_CnnLstm):
    """wakeword_training_script.py:141-165: three convs (1,32,64,128), LSTM(128,256,2,dropout .6)."""

    def __init__(self, config=ModelConfig, audio_config=AudioConfig):
        super().__init__([1, 32, 64, 128], config.HIDDEN_SIZE, config.NUM_LAYERS, config.DROPOUT, config.NUM_CLASSES)
        self.config = config
        self.audio_config = audio_config
        self.mel_height = audio_config.N_MELS
        self.mel_width = int(audio_config.SAMPLE_RATE * audio_config.DURATION / audio_config.HOP_LENGTH) + 1
        self.cnn_output_size = 128


def load_checkpoint(model: nn.Module, path: str, map_location=None):
    """Load a reference checkpoint dict ({'model_state_dict': ...}, wakeword_training_script.py:327-335 and
    :479-488, notebook cell 21) or a bare state_dict.  Uses `weights_only=True`: nothing in the file runs."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    model.load_state_dict(sd)
    return ckpt


def save_deployment_package(model: nn.Module, path: str, best_val_accuracy=0, epoch=0, device="cuda"):
    """Write the reference's deployment dict (notebook cell 21, wakeword_training.ipynb:951-977): model_state_dict,
    model_config, audio_config, training_info, classes.  Loadable by the reference and by `load_checkpoint`."""
    pkg = {
        "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
        "model_config": {"HIDDEN_SIZE": ModelConfig.HIDDEN_SIZE, "NUM_LAYERS": ModelConfig.NUM_LAYERS,
                         "DROPOUT": float(model.dropout.p), "NUM_CLASSES": ModelConfig.NUM_CLASSES},
        "audio_config": {k: getattr(AudioConfig, k) for k in ("SAMPLE_RATE", "DURATION", "N_MELS", "N_FFT", "HOP_LENGTH", "FMIN", "FMAX")},
        "training_info": {"best_val_accuracy": best_val_accuracy, "epoch": epoch, "device": str(device)},
        "classes": ["negative", "wakeword"],
    }
    torch.save(pkg, path)
    return pkg
