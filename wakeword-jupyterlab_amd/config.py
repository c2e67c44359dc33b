"""Constants of the path, mirroring the reference's config namespaces.

AudioConfig / ModelConfig : /root/reference/wakeword_training_script.py:29-43 (dup notebook cell 3)
Config                    : /root/reference/wakeword_training/train_wakeword.py:16-25
AugmentationConfig        : /root/reference/wakeword_training_script.py:52-58
Only the fields the accelerated path reads are kept (training hyper-parameters are out of scope).
"""


class AudioConfig:
    SAMPLE_RATE = 16000
    DURATION = 1.0
    N_MELS = 80
    N_FFT = 2048
    HOP_LENGTH = 512
    WIN_LENGTH = 2048
    FMIN = 0
    FMAX = 8000


class AugmentationConfig:
    AUGMENTATION_PROB = 0.8
    NOISE_FACTOR = 0.15
    TIME_SHIFT_MAX = 0.3
    PITCH_SHIFT_MAX = 3
    SPEED_CHANGE_MIN = 0.7
    SPEED_CHANGE_MAX = 1.3


class ModelConfig:            # WakewordModel (3 convs)
    HIDDEN_SIZE = 256
    NUM_LAYERS = 2
    DROPOUT = 0.6
    NUM_CLASSES = 2


class Config:                 # SimpleWakewordModel (2 convs)
    SAMPLE_RATE = 16000
    DURATION = 1.0
    N_MELS = 80
    HIDDEN_SIZE = 256
    NUM_LAYERS = 2
    DROPOUT = 0.5


CLIP_SAMPLES = int(AudioConfig.SAMPLE_RATE * AudioConfig.DURATION)          # 16000
N_FRAMES = 1 + CLIP_SAMPLES // AudioConfig.HOP_LENGTH                       # 32


def check_audio_config(cfg) -> None:
    """The kernels are built for exactly the reference constants; refuse anything else loudly."""
    want = {k: getattr(AudioConfig, k) for k in ("SAMPLE_RATE", "N_MELS", "N_FFT", "HOP_LENGTH", "WIN_LENGTH", "FMIN", "FMAX")}
    got = {k: getattr(cfg, k, None) for k in want}
    if float(getattr(cfg, "DURATION", 1.0)) != 1.0 or any(float(got[k]) != float(want[k]) for k in want):
        raise NotImplementedError(f"the HIP front-end is built for {want} at DURATION 1.0; got {got}")
