#!/usr/bin/env python3
"""The flow of the reference's `main()` (wakeword_training_script.py:395-495) on the MI355X path: sample data -> file lists -> split ->
AudioProcessor / WakewordModel -> WakewordDataset x 3 (training split augmented) -> DataLoader(..., num_workers=2) -> train / validate
epochs (the loop bodies of WakewordTrainer, :241-289) -> best / final checkpoints.  Only the imports differ from the reference's script:
every class and `DataLoader` come from `wakeword_jupyterlab_amd`; criterion, optimiser and scheduler are torch's.

    PYTHONPATH=. python examples/train_from_files.py [--epochs 10] [--data DIR] [--lr 1e-4]
"""
import argparse
import glob
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.optim as optim  # noqa: E402

from wakeword_jupyterlab_amd import AudioProcessor, DataLoader, WakewordDataset, WakewordModel  # noqa: E402
from wakeword_jupyterlab_amd.synth import create_sample_data  # noqa: E402


def split(files, test=0.1, val=0.2, seed=42):
    """train_test_split twice, as :439-443 (deterministic shuffle; sklearn is not needed for this)."""
    files = sorted(files)
    random.Random(seed).shuffle(files)
    n_test = max(1, int(round(len(files) * test)))
    rest = files[n_test:]
    n_val = max(1, int(round(len(rest) * val)))
    return rest[n_val:], rest[:n_val], files[:n_test]


def run_epoch(model, loader, criterion, device, optimizer=None):
    train = optimizer is not None
    model.train() if train else model.eval()
    loss_sum, correct, total = 0.0, 0, 0
    with torch.enable_grad() if train else torch.no_grad():
        for data, target in loader:
            data, target = data.to(device), target.to(device).squeeze()
            if train:
                optimizer.zero_grad()
            output = model(data)
            loss = criterion(output, target)
            if train:
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
                loss.backward()
                optimizer.step()
            loss_sum += loss.item()
            total += target.size(0)
            correct += (torch.max(output.data, 1)[1] == target).sum().item()
    return loss_sum / len(loader), 100.0 * correct / total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--data", default=".")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--batch-size", type=int, default=16)
    a = ap.parse_args()
    device = torch.device("cuda")
    print(f"Using device: {device} ({torch.cuda.get_device_name(0)})")
    wdir, ndir = os.path.join(a.data, "wakeword_data"), os.path.join(a.data, "negative_data")
    if not os.path.exists(wdir) or len(os.listdir(wdir)) == 0:
        create_sample_data(a.data)
    wake = [f for ext in ("*.wav", "*.mp3", "*.flac") for f in glob.glob(os.path.join(wdir, ext))]
    neg = [f for ext in ("*.wav", "*.mp3", "*.flac") for f in glob.glob(os.path.join(ndir, ext))]
    print(f"Wakeword files: {len(wake)}   Negative files: {len(neg)}")
    w_tr, w_va, w_te = split(wake)
    n_tr, n_va, n_te = split(neg)
    processor = AudioProcessor()
    model = WakewordModel().to(device)
    print(f"Parameters: {sum(p.numel() for p in model.parameters()):,}")
    train_loader = DataLoader(WakewordDataset(w_tr, n_tr, processor, augment=True), batch_size=a.batch_size, shuffle=True, num_workers=2)
    val_loader = DataLoader(WakewordDataset(w_va, n_va, processor, augment=False), batch_size=a.batch_size, shuffle=False, num_workers=2)
    test_loader = DataLoader(WakewordDataset(w_te, n_te, processor, augment=False), batch_size=a.batch_size, shuffle=False, num_workers=2)
    criterion = nn.CrossEntropyLoss().to(device)
    optimizer = optim.Adam(model.parameters(), lr=a.lr, weight_decay=1e-5)
    scheduler = optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="max", factor=0.5, patience=5)
    best = 0.0
    for epoch in range(a.epochs):
        tl, ta = run_epoch(model, train_loader, criterion, device, optimizer)
        vl, va = run_epoch(model, val_loader, criterion, device)
        scheduler.step(va)
        print(f"Epoch {epoch + 1}/{a.epochs}  Train Loss: {tl:.4f}, Train Acc: {ta:.2f}%  Val Loss: {vl:.4f}, Val Acc: {va:.2f}%")
        if va > best:
            best = va
            torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                        "val_acc": va, "train_acc": ta, "train_loss": tl, "val_loss": vl}, os.path.join(a.data, "best_wakeword_model.pth"))
    _, test_acc = run_epoch(model, test_loader, criterion, device)
    print(f"Best validation accuracy: {best:.2f}%   Test accuracy: {test_acc:.2f}%")
    torch.save({"model_state_dict": model.state_dict(), "best_val_acc": best, "device": str(device)}, os.path.join(a.data, "final_wakeword_model.pth"))


if __name__ == "__main__":
    main()
