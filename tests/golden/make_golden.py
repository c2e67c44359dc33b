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
  * the 3-conv `WakewordModel` (/root/reference/wakeword_training_script.py:141-184): its FILE imports
    librosa/soundfile/seaborn at top level (absent here: ordinary ModuleNotFoundError), but the CLASS needs only
    torch.  The file is parsed with `ast`, the class nodes `AudioConfig` (:29-37), `ModelConfig` (:39-43) and
    `WakewordModel` (:141-184) are compiled from the reference's own text and executed in a fresh namespace that
    holds the real `torch`, `nn`, `F` -- nothing else of the file runs, nothing of it is written anywhere.
    Weights = synth.make_state_dict('full', seed=1234); same two inputs; pooled (hook on `model.pool`) + logits
    -> model_full_seed1234.npz.
  * gradients of `SimpleWakewordModel` for the training path (wakeword_training_script.py:241-267: CrossEntropyLoss,
    loss.backward()): eval-mode forward (dropout = identity, so the numbers do not depend on torch's RNG stream),
    loss and every parameter's .grad for x32[:4] with labels [1,0,1,0] -> grads_simple_seed1234.npz; the same for the 3-conv
    WakewordModel class -> grads_full_seed1234.npz.
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

REF_SCRIPT = "/root/reference/wakeword_training_script.py"


def reference_full_model_classes():
    """The reference's own AudioConfig / ModelConfig / WakewordModel class statements, compiled from its text."""
    import ast

    import torch.nn as nn
    import torch.nn.functional as F
    tree = ast.parse(open(REF_SCRIPT).read(), REF_SCRIPT)
    want = ("AudioConfig", "ModelConfig", "WakewordModel")
    nodes = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in want]
    assert [n.name for n in nodes] == list(want), [n.name for n in nodes]
    ns = {"torch": torch, "nn": nn, "F": F, "__name__": "reference_classes"}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), REF_SCRIPT, "exec"), ns)
    return ns

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

    # ---- gradients of the same model (training path): CE loss, eval-mode forward so dropout is the identity ----
    gx = torch.from_numpy(x32[:4])
    labels = torch.tensor([1, 0, 1, 0])
    model.zero_grad()
    loss = torch.nn.CrossEntropyLoss()(model(gx), labels)
    loss.backward()
    grads = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}
    outg = os.path.join(HERE, "grads_simple_seed1234.npz")
    np.savez_compressed(outg, x=x32[:4], labels=labels.numpy(), loss=np.float32(loss.item()), **{"grad." + k: v for k, v in grads.items()})
    print("wrote", outg, "loss", loss.item(), {k: float(np.abs(v).max()) for k, v in grads.items()})

    # ---- the 3-conv WakewordModel, from the reference's own class text ----
    ns = reference_full_model_classes()
    sdf = synth.make_state_dict("full", seed=1234)
    full = ns["WakewordModel"]()
    full.load_state_dict({k: torch.from_numpy(v) for k, v in sdf.items()})
    full.eval()
    assert sum(p.numel() for p in full.parameters()) == 1014786          # model_architecture.txt:10
    full.pool.register_forward_hook(lambda m, i, o: pooled.__setitem__("v", o.detach().flatten(1).numpy().copy()))
    x32f = x32[:4]
    with torch.no_grad():
        yf32 = full(torch.from_numpy(x32f)).numpy()
    pf32 = pooled["v"]
    with torch.no_grad():
        yf31 = full(torch.from_numpy(x31)).numpy()
    pf31 = pooled["v"]
    outf = os.path.join(HERE, "model_full_seed1234.npz")
    np.savez_compressed(outf, x32=x32f, pooled32=pf32, logits32=yf32, x31=x31, pooled31=pf31, logits31=yf31,
                        weight_seed=np.int64(1234))
    print("wrote", outf, {k: v.shape for k, v in np.load(outf).items()})
    print("full logits32[:2] =", yf32[:2])

    # ---- gradients of the 3-conv model's own class (training path, WakewordTrainer.train_epoch :241-267), dropout = identity ----
    full.zero_grad()
    lossf = torch.nn.CrossEntropyLoss()(full(torch.from_numpy(x32f)), labels)
    lossf.backward()
    gradsf = {k: p.grad.detach().numpy().copy() for k, p in full.named_parameters()}
    # weight_hh gradients are exactly zero (h0 = 0): 2 x 1 MB of zeros that the test asserts instead of storing
    assert not gradsf["lstm.weight_hh_l0"].any() and not gradsf["lstm.weight_hh_l1"].any()
    outgf = os.path.join(HERE, "grads_full_seed1234.npz")
    np.savez_compressed(outgf, x=x32f, labels=labels.numpy(), loss=np.float32(lossf.item()),
                        **{"grad." + k: v for k, v in gradsf.items() if "weight_hh" not in k})
    print("wrote", outgf, "loss", lossf.item(), {k: float(np.abs(v).max()) for k, v in gradsf.items()})


if __name__ == "__main__":
    main()
