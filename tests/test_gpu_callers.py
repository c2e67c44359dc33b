"""The callers the drop-in serves, driven end to end on the GPU (SURVEY.md section 8 rows A12, B3, (f).4 and configs[4]):

  * the batched eval loop (notebook cell 17, wakeword_training.ipynb:727+; WakewordTrainer.validate,
    wakeword_training_script.py:269-289) over DataLoader(WakewordDataset(...), batch_size=16, num_workers=0)
  * predict_wakeword on a written WAV (notebook cell 19, wakeword_training.ipynb:871-893)
  * load_checkpoint of a {'model_state_dict': ...} file (wakeword_training_script.py:327-335) onto the device, checked
    against the reference-generated goldens
  * sharded_forward_pcm with two ranks (fresh processes, gloo rendezvous, both on the one GPU of the box)
  * the RCCL gather path (one-rank `nccl` group in a fresh process): the async double-buffered all-gather of bench.py's step
  * streaming at BASELINE configs[4]'s size: 256 microphones, 10 ms hop
"""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest
import torch

import wakeword_jupyterlab_amd as pkg
from oracle import mel_oracle, model_oracle
from wakeword_jupyterlab_amd import inference
from wakeword_jupyterlab_amd.audio import AudioProcessor
from wakeword_jupyterlab_amd.dataset import WakewordDataset
from wakeword_jupyterlab_amd.model import load_checkpoint, save_deployment_package

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOGIT_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _model(arch, sd, dev):
    m = pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev).eval()


def _write_wav16(path, x, sr=16000):
    raw = np.clip(np.round(np.asarray(x, np.float64) * 32767), -32768, 32767).astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16) \
        + b"data" + struct.pack("<I", len(raw))
    with open(path, "wb") as f:
        f.write(hdr + raw)


def _file_oracle(x16):
    """What the reference computes for a 16 kHz s16 file: decode (int16 / 32768) -> normalize -> pad -> log-mel."""
    a = (np.clip(np.round(np.asarray(x16, np.float64) * 32767), -32768, 32767).astype(np.int16).astype(np.float32) / 32768.0)
    a = a / np.abs(a).max()
    a = np.pad(a, (0, 16000 - len(a))) if len(a) < 16000 else a
    return mel_oracle.logmel_batch(a[None].astype(np.float32), normalize=False)[0]


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("arch", ["simple", "full"])
def test_eval_loop_over_dataloader_and_predict_wakeword(dev, tmp_path, arch):
    n_pos, n_neg = 13, 24                      # 37 files: the last DataLoader batch is ragged (37 = 2 * 16 + 5)
    clips = pkg.synth.make_clips(500, n_pos + n_neg) * 0.9
    clips[5, 9000:] = 0.0
    lens = [16000] * (n_pos + n_neg)
    lens[3], lens[20] = 7000, 12345            # short files: right zero-padded like pad_or_truncate
    paths = []
    for i in range(n_pos + n_neg):
        p = os.path.join(tmp_path, f"{'wake' if i < n_pos else 'neg'}_{i:03d}.wav")
        _write_wav16(p, clips[i, :lens[i]])
        paths.append(p)
    proc = AudioProcessor()
    ds = WakewordDataset(paths[:n_pos], paths[n_pos:], proc, augment=False, verbose=False)
    loader = torch.utils.data.DataLoader(ds, batch_size=16, shuffle=False, num_workers=0)
    sd = pkg.synth.make_state_dict(arch, seed=77)
    sd["fc.bias"] = np.array([0.0, 0.01], np.float32)
    m = _model(arch, sd, dev)

    ref_mel = np.stack([_file_oracle(clips[i, :lens[i]]) for i in range(n_pos + n_neg)])
    ref_logits = model_oracle.forward_np(ref_mel, sd)
    ref_pred = ref_logits.argmax(axis=1)
    labels = np.array([1] * n_pos + [0] * n_neg)

    # the items themselves
    x0, y0 = ds[3]
    assert x0.shape == (1, 80, 32) and x0.dtype == torch.float32 and y0.tolist() == [1]
    assert np.abs(x0.numpy() - ref_mel[3]).max() <= 1e-4

    crit = torch.nn.CrossEntropyLoss()
    preds, labs, loss, acc = inference.evaluate(m, loader, dev, criterion=crit)
    assert len(preds) == n_pos + n_neg and np.array_equal(np.asarray(labs), labels)
    margin = np.abs(ref_logits[:, 1] - ref_logits[:, 0])
    sure = margin > 2 * LOGIT_TOL
    assert np.array_equal(np.asarray(preds)[sure], ref_pred[sure])
    assert abs(acc - 100.0 * (ref_pred == labels).mean()) <= 100.0 * (~sure).sum() / len(labels) + 1e-9
    # mean of per-batch CE losses, as validate() computes it (:283-287)
    lt = torch.from_numpy(ref_logits)
    ref_loss = np.mean([crit(lt[s:s + 16], torch.from_numpy(labels[s:s + 16])).item() for s in range(0, len(labels), 16)])
    assert abs(loss - ref_loss) <= 1e-3
    preds2, labs2 = inference.evaluate(m, loader, dev)                         # notebook cell 17 form
    assert list(preds2) == list(preds) and list(labs2) == list(labs)

    # DataLoader workers cannot run the GPU-backed __getitem__: a clear error, not a HIP re-initialisation crash
    bad_loader = torch.utils.data.DataLoader(ds, batch_size=4, num_workers=1)
    with pytest.raises(RuntimeError, match="num_workers=0"):
        next(iter(bad_loader))

    # batched path (GPU decode) serves the same items
    got = torch.cat([d for d, _ in ds.batches(batch_size=16)]).cpu().numpy()
    assert np.abs(got[:, 0] - ref_mel[:, 0]).max() <= 1e-4

    # predict_wakeword: softmax(out)[0][1] >= threshold
    for i in (0, 3, 20):
        is_ww, p = inference.predict_wakeword(paths[i], m, proc, dev, threshold=0.5)
        e = np.exp(ref_logits[i] - ref_logits[i].max())
        ref_p = e[1] / e.sum()
        assert abs(p - ref_p) <= 1e-3 and (is_ww == (ref_p >= 0.5) or abs(ref_p - 0.5) < 1e-3)
    junk = os.path.join(tmp_path, "junk.mp3")
    open(junk, "wb").write(b"not audio")
    assert inference.predict_wakeword(junk, m, proc, dev) == (False, 0.0)     # reference: print + (False, 0.0)
    ds_bad = WakewordDataset([junk], [paths[0]], proc, verbose=False)
    xb, _ = ds_bad[0]
    assert ds_bad.unreadable == 1 and not xb.any() and xb.shape == (1, 80, 32)


def test_evaluate_pcm_matches_oracle(dev):
    clips = pkg.synth.make_clips(900, 70)
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = _model("simple", sd, dev)
    logits, pred = inference.evaluate_pcm(m, torch.from_numpy(clips).to(dev), batch_size=32)
    ref = model_oracle.forward_np(mel_oracle.logmel_batch(clips, normalize=True), sd)
    assert logits.shape == (70, 2) and pred.shape == (70,)
    assert np.abs(logits.cpu().numpy() - ref).max() <= LOGIT_TOL


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("arch,fixture", [("simple", "golden_simple"), ("full", "golden_full")])
def test_load_checkpoint_onto_the_device_reproduces_reference_logits(dev, tmp_path, request, arch, fixture):
    golden = request.getfixturevalue(fixture)
    sd = pkg.synth.make_state_dict(arch, seed=1234)
    # the reference's checkpoint dict (wakeword_training_script.py:327-335)
    ckpt = {"epoch": 3, "model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}, "optimizer_state_dict": {},
            "best_val_acc": 91.5, "train_losses": [0.7, 0.5], "val_losses": [0.6, 0.55]}
    path = os.path.join(tmp_path, "best_model.pth")
    torch.save(ckpt, path)
    m = (pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel()).to(dev).eval()     # constructed on the device first
    x = torch.from_numpy(golden["x32"]).to(dev)
    with torch.no_grad():
        before = m(x).cpu().numpy()
    info = load_checkpoint(m, path, map_location=dev)
    assert info["epoch"] == 3 and next(m.parameters()).device.type == "cuda"
    with torch.no_grad():
        after = m(x).cpu().numpy()
        after31 = m(torch.from_numpy(golden["x31"]).to(dev)).cpu().numpy()
    assert np.abs(before - golden["logits32"]).max() > 1e-3                     # random init differs: the load did something
    assert np.abs(after - golden["logits32"]).max() <= LOGIT_TOL               # packed weights were rebuilt from the file
    assert np.abs(after31 - golden["logits31"]).max() <= LOGIT_TOL
    # deployment package round trip (notebook cell 21) and a bare state_dict file
    pkg_path = os.path.join(tmp_path, "wakeword_deployment.pth")
    save_deployment_package(m, pkg_path, best_val_accuracy=91.5, epoch=3)
    m2 = (pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel()).to(dev).eval()
    load_checkpoint(m2, pkg_path, map_location="cpu")
    bare = os.path.join(tmp_path, "bare.pth")
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, bare)
    m3 = (pkg.SimpleWakewordModel() if arch == "simple" else pkg.WakewordModel()).to(dev).eval()
    load_checkpoint(m3, bare)
    with torch.no_grad():
        assert torch.equal(m2(x), m(x)) and torch.equal(m3(x), m(x))


# ---------------------------------------------------------------------------------------------------------------
def test_sharded_forward_pcm_two_ranks_on_one_gpu(tmp_path):
    """BASELINE configs[3]'s data path at world size 2: each rank runs PCM -> logits on its shard, the logits are gathered on every
    rank (gloo here -- host-staged -- because both ranks share the one GPU of this box; RCCL on the 8-GPU node) and must
    equal, clip for clip, what one process computes for the whole batch.  Ragged: 301 clips -> shards of 151 and 150."""
    port = 29500 + os.getpid() % 2000
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    child = os.path.join(ROOT, "tests", "_sharded_child.py")
    procs = [subprocess.Popen([sys.executable, child, str(301)], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{se[-3000:]}"
        assert "SHARDED_OK" in so, so[-2000:]


def test_shuffling_loader_is_the_dataloader_line_without_workers(dev, tmp_path):
    """`train_loader = DataLoader(train_dataset, batch_size=16, shuffle=True, num_workers=2)` (wakeword_training_script.py:461-463)
    -> `train_loader = train_dataset.loader(batch_size=16, shuffle=True)`: same shapes per batch, every item exactly once per epoch, a new
    order each epoch, and every (data, target) pair still belongs together (checked against the per-file oracle by content)."""
    n = 37
    paths, mels = [], []
    for i in range(n):
        x = pkg.synth.make_clip(700 + i)[: 16000 - 211 * (i % 7)] * 0.4
        p = os.path.join(tmp_path, f"s{i:02d}.wav")
        _write_wav16(p, x)
        paths.append(p)
        mels.append(_file_oracle(x))
    ref = np.stack(mels)                                   # [n, 1?, 80, 32] by file index
    ref = ref.reshape(n, 80, 32)
    wake, neg = paths[:15], paths[15:]
    ds = WakewordDataset(wake, neg, AudioProcessor(), verbose=False)
    from wakeword_jupyterlab_amd import DataLoader                       # the reference's own line, only the import differs
    ld = DataLoader(ds, batch_size=16, shuffle=True, num_workers=2)
    assert len(ld) == 3 and type(ld) is type(ds.loader(batch_size=16, shuffle=True))
    orders = []
    for epoch in range(2):
        seen = []
        for data, target in ld:
            assert data.is_cuda and data.shape[1:] == (1, 80, 32) and target.shape == (data.shape[0], 1) and data.shape[0] in (16, 5)
            got = data[:, 0].cpu().numpy()
            for row, lab in zip(got, target[:, 0].tolist()):
                j = int(np.argmin(np.abs(ref - row[None]).max(axis=(1, 2))))
                assert np.abs(ref[j] - row).max() <= 1e-4 and lab == (1 if j < 15 else 0)
                seen.append(j)
        assert sorted(seen) == list(range(n))
        orders.append(seen)
    assert orders[0] != orders[1] and orders[0] != list(range(n))
    assert [int(t) for _, tg in ds.loader(16) for t in tg[:, 0]] == ds.labels           # unshuffled: file order
    # a file too long for the staging buffer in the middle of an epoch: the loader regrows the reader and goes on from that batch
    long = os.path.join(tmp_path, "long.wav")
    _write_wav16(long, np.tile(pkg.synth.make_clip(5) * 0.3, 400))                        # 400 s = 12.8 MB of samples
    ds2 = WakewordDataset(paths[:6] + [long] + paths[6:12], [], AudioProcessor(), verbose=False)
    got = [d for d, _ in ds2.loader(batch_size=4)]
    assert [g.shape[0] for g in got] == [4, 4, 4, 1] and all(torch.isfinite(g).all() for g in got)
    flat = torch.cat(got)[:, 0].cpu().numpy()
    for k, j in enumerate(list(range(6)) + [None] + list(range(6, 12))):
        if j is not None:
            assert np.abs(flat[k] - ref[j]).max() <= 1e-4


def test_rccl_gather_path_one_rank():
    """The RCCL path itself (`backend="nccl"`), executed: a fresh process with a one-rank RCCL group on this box's GPU runs the
    double-buffered async all-gather of bench.py's step (distributed.LogitsGatherPipeline) for seven steps, `sharded_forward_pcm`
    and `all_gather_logits`; gathered logits must equal the local ones bit for bit.  The 8-GPU run is then not the first time
    this code meets RCCL."""
    port = 31500 + os.getpid() % 2000
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_child.py")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "RCCL_OK" in p.stdout, p.stdout[-2000:]


# ---------------------------------------------------------------------------------------------------------------
def test_streaming_256_microphones_10ms_hop():
    """BASELINE configs[4] at its stated size: 256 concurrent microphones, 10 ms hop, the per-hop step replayed from its hipGraph.
    130 hops (the window fills after 100); every 10th hop the logits of 16 microphones spread over the batch are checked against
    the windowed oracle, and every microphone's probability is checked for consistency with its logits."""
    dev = torch.device("cuda", 0)
    n_mics, hop, n_hops = 256, 160, 130
    sd = pkg.synth.make_state_dict("simple", seed=1234)
    m = _model("simple", sd, dev)
    total = n_hops * hop
    base = np.stack([np.concatenate([pkg.synth.make_clip(40 * i + j) for j in range(total // 16000 + 1)])[:total] for i in range(32)])
    gains = (1.0 + np.arange(n_mics) / n_mics).astype(np.float32)
    streams = (np.roll(np.tile(base, (n_mics // 32, 1)), 0, axis=0) * gains[:, None]).astype(np.float32)
    for i in range(n_mics):                                  # every microphone its own signal: rotate by a mic-specific lag
        streams[i] = np.roll(streams[i], 37 * i)
    dev_streams = torch.from_numpy(streams).to(dev)
    det = pkg.StreamingDetector(m, n_mics=n_mics, hop_samples=hop, threshold=0.5)
    subset = np.arange(7, n_mics, 16)                        # 16 microphones
    checked = 0
    for k in range(n_hops):
        prob = det.step(dev_streams[:, k * hop:(k + 1) * hop])
        if k % 10 == 9 or k == n_hops - 1:
            det.stream.synchronize()
            done = (k + 1) * hop
            win = np.zeros((n_mics, 16000), np.float32)
            seg = streams[:, max(0, done - 16000):done]
            win[:, 16000 - seg.shape[1]:] = seg              # before the window fills, the oldest samples are zeros
            if k in (9, 99, n_hops - 1):
                assert np.array_equal(det.window().cpu().numpy(), win)
            ref = model_oracle.forward_np(mel_oracle.logmel_batch(win[subset], normalize=True), sd)
            got = det.logits.cpu().numpy()
            assert np.abs(got[subset] - ref).max() <= LOGIT_TOL, f"hop {k}"
            p = prob.cpu().numpy()
            e = np.exp(got - got.max(axis=1, keepdims=True))
            assert np.isfinite(p).all() and np.abs(p - e[:, 1] / e.sum(axis=1)).max() <= 1e-5
            checked += 1
    assert checked >= 13
    det.close()


def test_the_reference_main_flow_with_only_the_imports_changed(dev, tmp_path, capsys):
    """`main()` of wakeword_training_script.py:395-495 with the package's names in place of the script's class definitions: create_sample_data
    -> file lists -> split -> AudioProcessor / WakewordModel().to(device) -> three WakewordDatasets (training split augmented) -> the
    DataLoader(..., num_workers=2) lines as written -> WakewordTrainer's train_epoch / validate loop bodies (:241-289, restated here: the
    reference cannot travel) -> checkpoint dict -> load_checkpoint.  The synthetic classes (tone + noise vs noise) are separable: a few
    epochs must learn them."""
    import glob
    import random
    from wakeword_jupyterlab_amd import DataLoader, WakewordModel
    from wakeword_jupyterlab_amd.synth import create_sample_data
    np.random.seed(0); random.seed(0); torch.manual_seed(0)
    create_sample_data(str(tmp_path))
    wake = sorted(glob.glob(os.path.join(tmp_path, "wakeword_data", "*.wav")))
    neg = sorted(glob.glob(os.path.join(tmp_path, "negative_data", "*.wav")))
    assert (len(wake), len(neg)) == (50, 100)
    rng = random.Random(42)
    rng.shuffle(wake); rng.shuffle(neg)
    w_tr, w_va, w_te = wake[:36], wake[36:45], wake[45:]
    n_tr, n_va, n_te = neg[:72], neg[72:90], neg[90:]
    processor = AudioProcessor()
    model = WakewordModel().to(dev)
    train_loader = DataLoader(WakewordDataset(w_tr, n_tr, processor, augment=True, verbose=False), batch_size=16, shuffle=True, num_workers=2)
    val_loader = DataLoader(WakewordDataset(w_va, n_va, processor, augment=False, verbose=False), batch_size=16, shuffle=False, num_workers=2)
    test_loader = DataLoader(WakewordDataset(w_te, n_te, processor, augment=False, verbose=False), batch_size=16, shuffle=False, num_workers=2)
    criterion = torch.nn.CrossEntropyLoss().to(dev)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)

    def run(loader, train):
        model.train() if train else model.eval()
        loss_sum, correct, total = 0.0, 0, 0
        with torch.enable_grad() if train else torch.no_grad():
            for data, target in loader:
                data, target = data.to(dev), target.to(dev).squeeze()
                if train:
                    optimizer.zero_grad()
                output = model(data)
                loss = criterion(output, target)
                if train:
                    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
                    loss.backward()
                    optimizer.step()
                loss_sum += loss.item()
                _, predicted = torch.max(output.data, 1)
                total += target.size(0)
                correct += (predicted == target).sum().item()
        return loss_sum / len(loader), 100.0 * correct / total

    history = []
    for epoch in range(4):
        tl, ta = run(train_loader, True)
        vl, va = run(val_loader, False)
        history.append((tl, ta, vl, va))
    print("history", history)
    assert all(np.isfinite(h).all() for h in history)
    assert history[-1][0] < history[0][0] and history[-1][3] >= 90.0, history
    _, test_acc = run(test_loader, False)
    assert test_acc >= 90.0, test_acc
    path = os.path.join(tmp_path, "best_wakeword_model.pth")
    torch.save({"epoch": 3, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(), "val_acc": history[-1][3]}, path)
    m2 = WakewordModel().to(dev)
    ckpt = load_checkpoint(m2, path, map_location=dev)
    assert ckpt["epoch"] == 3
    m2.eval(); model.eval()
    data, _ = next(iter(test_loader))
    with torch.no_grad():
        assert torch.equal(m2(data), model(data))
