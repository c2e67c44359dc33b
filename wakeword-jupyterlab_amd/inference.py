"""The callers the drop-in serves: the batched eval loop and single-file predict.

  evaluate(model, loader, device)   <- notebook cell 17 lines 8-19 (wakeword_training.ipynb:727+) and
                                       WakewordTrainer.validate (wakeword_training_script.py:269-289)
  predict_wakeword(path, model, processor, device, threshold=0.8)
                                    <- notebook cell 19 (wakeword_training.ipynb:871-893)
  evaluate_pcm(model, pcm, batch)   the same loop fed with PCM already in HBM (bench / multi-GPU path)
"""
from __future__ import annotations

import numpy as np
import torch


def evaluate(model, loader, device, criterion=None):
    """Returns (all_preds, all_labels[, mean loss, accuracy %]).  `loader` yields (data [B,1,80,T], target [B,1])."""
    model.eval()
    all_preds, all_labels = [], []
    total_loss, correct, total, n_batches = 0.0, 0, 0, 0
    with torch.no_grad():
        for data, target in loader:
            data, target = data.to(device), target.to(device).squeeze()
            output = model(data)
            if criterion is not None:
                total_loss += criterion(output, target.reshape(-1)).item()
            _, predicted = torch.max(output, 1)
            total += target.numel()
            correct += (predicted == target.reshape(-1)).sum().item()
            n_batches += 1
            all_preds.extend(predicted.cpu().numpy())
            all_labels.extend(np.atleast_1d(target.cpu().numpy()))
    if criterion is None:
        return all_preds, all_labels
    return all_preds, all_labels, total_loss / max(1, n_batches), 100.0 * correct / max(1, total)


def predict_wakeword(audio_file_path, model, processor, device, threshold=0.8):
    """Single file -> (is_wakeword, probability); (False, 0.0) when the file cannot be processed."""
    model.eval()
    mel_spec = processor.process_audio_file(audio_file_path, augment=False)
    if mel_spec is None:
        print(f"Error processing audio file: {audio_file_path}")
        return False, 0.0
    mel_tensor = torch.FloatTensor(np.asarray(mel_spec, dtype=np.float32)).unsqueeze(0).unsqueeze(0).to(device)
    with torch.no_grad():
        output = model(mel_tensor)
        probabilities = torch.softmax(output, dim=1)
        wakeword_prob = probabilities[0][1].item()
    return wakeword_prob >= threshold, wakeword_prob


def evaluate_pcm(model, pcm: torch.Tensor, batch_size: int = 4096, normalize: bool = True):
    """PCM [N, n<=16000] on the device -> (logits [N,2], predictions [N]) in batches of `batch_size`."""
    model.eval()
    outs = []
    with torch.no_grad():
        for s in range(0, pcm.shape[0], batch_size):
            outs.append(model.forward_pcm(pcm[s:s + batch_size], normalize))
    logits = torch.cat(outs) if outs else torch.empty((0, 2), device=pcm.device)
    return logits, (torch.max(logits, 1)[1] if len(logits) else torch.empty(0, dtype=torch.long, device=pcm.device))
