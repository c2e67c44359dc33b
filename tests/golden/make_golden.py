#!/usr/bin/env python3
"""Generate tests/golden/model_simple_*.npz by running the REFERENCE module itself.

Runs only in the build container (needs /root/reference); the fixtures it writes are
plain data (inputs + outputs) and are what travels.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is pinned
  * `SimpleWakewordModel` (/root/reference/wakeword_training/train_wakeword.py:28-49), eval mode,
    weights = synth.make_state_dict('simple', seed=1234) loaded through `load_state_dict`:
      - x32 [8,1,80,32]: oracle log-mel of synthetic clips 0..7 (values in [-80, 0], the real input range)
      - x31 [4,1,80,31]: standard-normal inputs, the shape `SimpleDataset` feeds (train_wakeword.py:56)
    outputs: pooled features (hook on `model.pool`) and logits.
  * the 3-conv `WakewordModel` cannot be imported (its file imports librosa/soundfile/seaborn at
    top level, all absent), so it has no reference-generated fixture; see tests/test_oracle_model.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/wakeword_training")
sys.dont_write_bytecode = True

import train_wakeword as ref  # noqa: E402  (the reference, read-only)

import wakeword_jupyterlab_amd.synth as synth  # noqa: E402
from oracle import mel_oracle  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    sd = synth.make_state_dict("simple", seed=1234)
    model = ref.SimpleWakewordModel()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()

    pooled = {}
    model.pool.register_forward_hook(lambda m, i, o: pooled.__setitem__("v", o.detach().flatten(1).numpy().copy()))

    x32 = mel_oracle.logmel_batch(synth.make_clips(0, 8), normalize=True)
    with torch.no_grad():
        y32 = model(torch.from_numpy(x32)).numpy()
    p32 = pooled["v"]

    x31 = synth.normal(777, 4 * 80 * 31).astype(np.float32).reshape(4, 1, 80, 31)
    with torch.no_grad():
        y31 = model(torch.from_numpy(x31)).numpy()
    p31 = pooled["v"]

    out = os.path.join(HERE, "model_simple_seed1234.npz")
    np.savez_compressed(out, x32=x32, pooled32=p32, logits32=y32, x31=x31, pooled31=p31, logits31=y31,
                        weight_seed=np.int64(1234))
    print("wrote", out, {k: v.shape for k, v in np.load(out).items()})
    print("logits32[:2] =", y32[:2])


if __name__ == "__main__":
    main()
