"""CPU tests of the native batch WAV reader (csrc/ww_files.cpp) in its host-only mode: RIFF walk, threaded reads into the
staging buffer, status codes, capacity handling.  The same code path feeds pinned staging on the GPU box
(tests/test_gpu_decode.py drives it through K0 against oracle/decode_oracle.py).

Reference behaviour being replaced: AudioProcessor.load_audio, /root/reference/wakeword_training_script.py:65-71 (one file at a
time, print + None on failure); the expected header fields come from tests/wavio.py, the tests' own numpy parser.
"""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat
from wakeword_jupyterlab_amd import files
from wavio import _parse_wav


def _chunk(cid, body):
    return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")


def _wav(raw, sr=16000, bits=16, channels=1, tag=1, extra_before=b"", extra_after=b"", extensible=False, data_size=None):
    fmt = struct.pack("<HHIIHH", 0xFFFE if extensible else tag, channels, sr, sr * channels * bits // 8, channels * bits // 8, bits)
    if extensible:
        fmt += struct.pack("<HHI", 22, bits, 0) + struct.pack("<H", tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    data = b"data" + struct.pack("<I", len(raw) if data_size is None else data_size) + raw + (b"\0" if len(raw) & 1 else b"")
    body = b"WAVE" + _chunk(b"fmt ", fmt) + extra_before + data + extra_after
    return b"RIFF" + struct.pack("<I", len(body)) + body


@pytest.fixture()
def corpus(tmp_path):
    r = np.random.default_rng(7)
    def s16(n, ch=1): return r.integers(-30000, 30000, n * ch).astype("<i2").tobytes()
    cases = {
        "plain16.wav": _wav(s16(16000)),
        "short.wav": _wav(s16(777)),
        "stereo44k.wav": _wav(s16(5000, 2), sr=44100, channels=2),
        "list_before.wav": _wav(s16(4000), extra_before=_chunk(b"LIST", b"INFOISFT" + b"x" * 5001)),      # data chunk beyond the 4 KiB window
        "fact_after.wav": _wav(s16(1234), extra_after=_chunk(b"fact", b"\x01\x02\x03")),
        "u8.wav": _wav(r.integers(0, 255, 3001).astype(np.uint8).tobytes(), bits=8),
        "s24.wav": _wav(r.integers(0, 255, 3 * 2000).astype(np.uint8).tobytes(), bits=24, sr=48000),
        "s32.wav": _wav(r.integers(-2 ** 31, 2 ** 31 - 1, 900).astype("<i4").tobytes(), bits=32, sr=22050),
        "f32.wav": _wav(r.standard_normal(1500).astype("<f4").tobytes(), bits=32, tag=3, sr=8000),
        "ext.wav": _wav(s16(3000), extensible=True),
        "cut.wav": _wav(s16(2000), data_size=100000),                         # header promises more than the file holds
        "empty.wav": _wav(b""),
        "f64.wav": _wav(r.standard_normal(10).astype("<f8").tobytes(), bits=64, tag=3),
        # unusable
        "alaw.wav": _wav(r.integers(0, 255, 100).astype(np.uint8).tobytes(), bits=8, tag=6),
        "notwav.wav": b"this is not a wave file at all",
        "nodata.wav": b"RIFF" + struct.pack("<I", 4 + 24) + b"WAVE" + _chunk(b"fmt ", struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16)),
        "tiny.wav": b"RIFF",
    }
    paths = []
    for name, blob in cases.items():
        p = os.path.join(tmp_path, name)
        with open(p, "wb") as f:
            f.write(blob)
        paths.append(p)
    paths.append(os.path.join(tmp_path, "missing.wav"))
    return paths, cases


FMT_OF = {(1, 16): nat.FMT_S16, (1, 24): nat.FMT_S24, (1, 32): nat.FMT_S32, (3, 32): nat.FMT_F32, (1, 8): nat.FMT_U8, (3, 64): nat.FMT_F64}
EXPECT_BAD = {"alaw.wav": -4, "notwav.wav": -2, "nodata.wav": -3, "tiny.wav": -2, "missing.wav": -1}


@pytest.mark.parametrize("threads", [1, 4])
def test_reader_stages_every_file_like_the_python_parser(corpus, threads):
    paths, cases = corpus
    rd = files.WavBatchReader(max_clips=64, max_raw_bytes=1 << 20, threads=threads, slots=2, host_only=True)
    for slot in (0, 1, 0):                                                     # slots are reusable
        descs, status = rd.read(paths, slot)
        stage = rd.staging(slot)
        spans = []
        for p, d, st in zip(paths, descs, status):
            name = os.path.basename(p)
            if name in EXPECT_BAD:
                assert st == EXPECT_BAD[name], (name, st)
                assert d["n_frames"] == 0 and d["up"] == 1 and d["down"] == 1     # K0 sees an empty clip -> a zero row
                continue
            assert st == 1, (name, st)
            blob = cases[name]
            tag, ch, sr, bits, start, length = _parse_wav(blob)
            fb = ch * bits // 8
            assert (d["channels"], d["sample_rate"], d["format"], d["n_frames"]) == (ch, sr, FMT_OF[(tag, bits)], length // fb), name
            nbytes = int(d["n_frames"]) * fb
            off = int(d["byte_offset"])
            assert off % 16 == 0 and bytes(stage[off:off + nbytes]) == blob[start:start + nbytes], name
            assert not stage[off + nbytes:(off + nbytes + 15) // 16 * 16].any()      # the alignment pad is zeroed
            spans.append((off, (nbytes + 15) // 16 * 16))
            from math import gcd
            g = gcd(sr, 16000)
            assert (d["up"], d["down"]) == (16000 // g, sr // g) and d["crop_start"] == 0
        spans.sort()
        assert all(a[0] + a[1] <= b[0] for a, b in zip(spans, spans[1:]))            # the bump allocator never overlaps files
        assert sum(s[1] for s in spans) == len(stage)
    with pytest.raises(RuntimeError):
        rd.decode(0)                                                             # host-only: nothing to decode on
    rd.close()


def test_reader_reports_the_size_it_needs_and_load_regrows(corpus):
    paths, _ = corpus
    rd = files.WavBatchReader(max_clips=64, max_raw_bytes=4096, threads=3, host_only=True)
    with pytest.raises(nat.NativeError) as e:
        rd.read(paths, 0)
    assert e.value.code == nat.WW_ENOSPACE and e.value.needed > 4096
    need = e.value.needed
    rd._regrow(64, need)
    descs, status = rd.read(paths, 0)
    assert (status == 1).sum() == len(paths) - len(EXPECT_BAD) and len(rd.staging(0)) == need
    with pytest.raises(nat.NativeError):
        rd.read(paths * 5, 0)                                                    # more files than the reader was created for
    assert rd.read([], 1)[0].size == 0
    rd.close()


def test_size_fields_that_lie_cost_one_file_not_the_batch(tmp_path):
    """Round-3 review: sizes written INSIDE a file must not decide how much staging it gets.  A streaming header (RIFF and data size
    0xFFFFFFFF, what ffmpeg / sox write into a pipe) on a 160 KB file, a header promising 800 k frames on 100 KB of samples and a
    never-finalised header (data size 0) are each read to the end of the file, like libsndfile does, with the frame count that is really
    there -- and a batch that holds them fits a staging buffer sized for the bytes on disk (no WW_ENOSPACE, no 4 GB regrow)."""
    r = np.random.default_rng(11)
    def s16(n): return r.integers(-30000, 30000, n).astype("<i2").tobytes()
    def streaming(raw):
        blob = bytearray(_wav(raw, data_size=0xFFFFFFFF))
        blob[4:8] = struct.pack("<I", 0xFFFFFFFF)
        return bytes(blob)
    cases = {
        "stream_big.wav": streaming(s16(80000)),                                # 160 KB: beyond the 68 KB head window
        "stream_small.wav": streaming(s16(3000)),
        "promises_800k.wav": _wav(s16(50000), data_size=1600000),                # 100 KB of samples
        "unfinalised.wav": _wav(s16(20001), data_size=0),
        "unfinalised_big.wav": _wav(s16(70000), data_size=0),
        "odd_tail.wav": streaming(s16(4000)) + b"\x7f",                         # a trailing half frame is dropped
        "plain.wav": _wav(s16(16000)),
    }
    frames = {"stream_big.wav": 80000, "stream_small.wav": 3000, "promises_800k.wav": 50000, "unfinalised.wav": 20001,
              "unfinalised_big.wav": 70000, "odd_tail.wav": 4000, "plain.wav": 16000}
    paths = []
    for name, blob in cases.items():
        p = os.path.join(tmp_path, name)
        with open(p, "wb") as f:
            f.write(blob)
        paths.append(p)
    on_disk = sum((2 * n + 15) // 16 * 16 for n in frames.values())
    rd = files.WavBatchReader(max_clips=16, max_raw_bytes=on_disk, threads=3, slots=2, host_only=True)
    descs, status = rd.read(paths, 0)                                            # would raise WW_ENOSPACE if a header's size were believed
    stage = rd.staging(0)
    assert len(stage) == on_disk
    for p, d, st in zip(paths, descs, status):
        name = os.path.basename(p)
        assert st == 1 and int(d["n_frames"]) == frames[name], (name, st, int(d["n_frames"]))
        tag, ch, sr, bits, start, length = _parse_wav(cases[name])
        off, nbytes = int(d["byte_offset"]), 2 * frames[name]
        assert bytes(stage[off:off + nbytes]) == cases[name][start:start + nbytes], name
        info = files.probe(p)
        assert info["n_frames"] == frames[name] and info["data_offset"] == start, name
    rd.close()


def test_one_reader_serves_one_stream_at_a_time_and_draws_crops_on_the_callers_thread(tmp_path):
    """Round-3 review: (1) two loaders over ONE processor (the reference hands one AudioProcessor to its train / val / test datasets) shared
    one reader's slots and thread pool and corrupted each other's batches without a word: a second stream() / load() on a reader whose
    stream is still alive now raises, and works again once the first is exhausted or closed; (2) pad_or_truncate's crops were drawn with
    the global `random` on the helper thread, racing the caller's own draws: they are drawn on the caller's thread now, so two seeded
    passes over files longer than 1 s give the same crops, also with other draws of the caller between the batches."""
    import random
    import threading
    r = np.random.default_rng(3)
    paths = []
    for i in range(24):
        p = os.path.join(tmp_path, f"long{i:02d}.wav")
        with open(p, "wb") as f:
            f.write(_wav(r.integers(-30000, 30000, 16000 + 977 * (i + 1)).astype("<i2").tobytes()))
        paths.append(p)
    rd = files.WavBatchReader(max_clips=8, max_raw_bytes=1 << 20, threads=2, slots=3, host_only=True)
    first = rd.stream(paths, 8, verbose=False)
    next(first)
    with pytest.raises(RuntimeError, match="still active"):
        next(rd.stream(paths, 8, verbose=False))
    with pytest.raises(RuntimeError, match="still active"):
        rd.load(paths[:4])
    first.close()
    assert sum(1 for _ in rd.stream(paths, 8, verbose=False)) == 3           # free again

    def crops(seed):
        random.seed(seed)
        out, threads = [], set()
        orig = files.WavBatchReader.draw_crops
        def spy(descs, status, n=files.CLIP_SAMPLES):
            threads.add(threading.get_ident())
            orig(descs, status, n)
            out.extend(int(c) for c in descs["crop_start"])
        files.WavBatchReader.draw_crops = staticmethod(spy)
        try:
            for _slot, ok in rd.stream(paths, 8, verbose=False):
                assert ok.all()
                random.random(); random.random()                             # the caller's own draws (augmentation plans) between batches
        finally:
            files.WavBatchReader.draw_crops = staticmethod(orig)
        assert threads == {threading.get_ident()}                            # drawn on the consumer's thread
        return out
    a, b, c = crops(7), crops(7), crops(8)
    assert a == b and a != c and len(a) == 24 and all(x >= 0 for x in a) and max(a) > 0
    rd.close()


def test_many_files_many_threads_are_all_accounted_for(tmp_path):
    """512 files of different lengths through 8 threads, three rounds: every payload arrives once, bit for bit."""
    r = np.random.default_rng(1)
    paths, payloads = [], []
    for i in range(512):
        raw = r.integers(-2 ** 15, 2 ** 15 - 1, 50 + 13 * (i % 97)).astype("<i2").tobytes()
        p = os.path.join(tmp_path, f"c{i:03d}.wav")
        with open(p, "wb") as f:
            f.write(_wav(raw))
        paths.append(p); payloads.append(raw)
    rd = files.WavBatchReader(max_clips=512, max_raw_bytes=2 << 20, threads=8, slots=3, host_only=True)
    for rnd in range(3):
        descs, status = rd.read(paths, rnd)
        stage = rd.staging(rnd)
        assert (status == 1).all()
        for d, raw in zip(descs, payloads):
            assert bytes(stage[int(d["byte_offset"]):int(d["byte_offset"]) + len(raw)]) == raw
    rd.close()


def test_encoded_paths_window_reads_the_same_files(corpus):
    paths, _ = corpus
    enc = files.EncodedPaths(paths)
    rd = files.WavBatchReader(max_clips=64, max_raw_bytes=1 << 20, threads=3, host_only=True)
    d_all, s_all = rd.read(paths, 0)
    want = [(int(d["n_frames"]), int(d["sample_rate"]), int(st)) for d, st in zip(d_all, s_all)]
    for lo, hi in ((0, len(paths)), (3, 9), (len(paths) - 2, len(paths)), (5, 5)):
        d, st = rd.read(enc, 1, lo, hi)
        assert [(int(a["n_frames"]), int(a["sample_rate"]), int(b)) for a, b in zip(d, st)] == want[lo:hi]
    rd.close()


def test_stream_reads_one_batch_ahead_and_stops_cleanly(tmp_path):
    """stream(): the helper thread fills the next slot while the consumer holds the current one; every payload arrives in order; breaking
    out of the generator stops and joins the thread; a reader with two slots is refused."""
    r = np.random.default_rng(2)
    paths, payloads = [], []
    for i in range(130):
        raw = r.integers(-2 ** 15, 2 ** 15 - 1, 64 + 7 * (i % 31)).astype("<i2").tobytes()
        p = os.path.join(tmp_path, f"q{i:03d}.wav")
        with open(p, "wb") as f:
            f.write(_wav(raw))
        paths.append(p); payloads.append(raw)
    rd = files.WavBatchReader(max_clips=32, max_raw_bytes=1 << 18, threads=4, slots=3, host_only=True)
    seen = 0
    for slot, ok in rd.stream(paths, 32):
        stage = rd.staging(slot)                             # still intact: the producer is at most one slot ahead, in another slot
        descs = np.frombuffer((C.c_char * (len(ok) * files.DESC_DTYPE.itemsize)).from_address(
            C.addressof(C.cast(_descs_ptr(rd, slot), C.POINTER(nat.ClipDesc)).contents)), dtype=files.DESC_DTYPE, count=len(ok))
        assert ok.all()
        for d, raw in zip(descs, payloads[seen:seen + len(ok)]):
            assert bytes(stage[int(d["byte_offset"]):int(d["byte_offset"]) + len(raw)]) == raw
        seen += len(ok)
    assert seen == 130
    import threading
    gen = rd.stream(paths, 16)
    next(gen)
    gen.close()                                              # consumer walks away after one batch
    assert not any(t.name == "ww-wav-reader" and t.is_alive() for t in threading.enumerate())
    with pytest.raises(ValueError):
        next(files.WavBatchReader(max_clips=32, threads=2, slots=2, host_only=True).stream(paths, 16))
    rd.close()


def _descs_ptr(rd, slot):
    """The slot's descriptor array (the library hands it out on every read; ask again with an empty batch-preserving call)."""
    return rd._last_descs[slot]


def test_probe_reads_one_header(corpus):
    paths, cases = corpus
    for p in paths:
        name = os.path.basename(p)
        info = files.probe(p)
        if name in EXPECT_BAD:
            assert info is None
        else:
            tag, ch, sr, bits, start, length = _parse_wav(cases[name])
            assert info["data_offset"] == start and info["n_frames"] == length // (ch * bits // 8) and info["sample_rate"] == sr


def test_crop_draw_uses_python_random_like_pad_or_truncate():
    import random
    descs = np.zeros(3, dtype=files.DESC_DTYPE)
    descs["n_frames"], descs["up"], descs["down"] = [16000, 48000 * 2, 30000], [1, 1, 1], [1, 3, 1]     # 1 s, 2 s at 48 kHz, 1.875 s
    status = np.array([1, 1, -1], dtype=np.int8)
    random.seed(5); want = random.randint(0, 32000 - 16000); random.seed(5)
    files.WavBatchReader.draw_crops(descs, status)
    assert list(descs["crop_start"]) == [0, want, 0]


def test_reader_is_clean_under_thread_sanitizer(tmp_path):
    """The reader's own source built -fsanitize=thread (plain C++, CPU only) and driven by tests/c/reader_tsan_harness.cpp: 8 threads,
    3 slots, 30 rounds over 302 files (two of them unusable).  Any data race in the pool / bump allocator / status writes is a report."""
    import shutil
    import subprocess
    import torch
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no clang++ with a ThreadSanitizer runtime")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc, libdir = os.path.join(root, "wakeword-jupyterlab_amd", "csrc"), os.path.join(root, "wakeword-jupyterlab_amd")
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    obj, exe = os.path.join(tmp_path, "files_tsan.o"), os.path.join(tmp_path, "reader_tsan")
    common = [clang, "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I", os.path.join(root, "include")]
    subprocess.run(common + ["-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", csrc, "-DWW_BUILD", "-x", "c++", "-c",
                             os.path.join(csrc, "ww_files.cpp"), "-o", obj], check=True, capture_output=True, text=True)
    subprocess.run(common + [os.path.join(root, "tests", "c", "reader_tsan_harness.cpp"), obj, "-o", exe, "-L", libdir, "-l:libwakeword_amd.so",
                             "-Wl,-rpath," + libdir, "-L", tl, "-Wl,-rpath," + tl, "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"],
                   check=True, capture_output=True, text=True)
    r = np.random.default_rng(0)
    paths = []
    for i in range(300):
        p = os.path.join(tmp_path, f"t{i:03d}.wav")
        with open(p, "wb") as f:
            f.write(_wav(r.integers(-2 ** 15, 2 ** 15 - 1, 200 + 37 * (i % 53)).astype("<i2").tobytes()))
        paths.append(p)
    bad = os.path.join(tmp_path, "bad.wav")
    with open(bad, "wb") as f:
        f.write(b"junk")
    # the sanitizer tracks descriptors by NUMBER for the whole process: pool threads on private tables (each one's first file is its
        # descriptor 3) read to it as races on "file descriptor 3" -- the harness keeps them on the shared table
        env = dict(os.environ, WW_READER_SHARED_FDS="1", TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66", LD_LIBRARY_PATH=tl + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe] + paths + [bad, os.path.join(tmp_path, "missing.wav")], capture_output=True, text=True, env=env, timeout=300)
    assert "WARNING: ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
    assert out.returncode == 0 and "READER_TSAN_OK 9000" in out.stdout, (out.returncode, out.stdout[-500:], out.stderr[-1500:])


def test_pool_threads_read_on_descriptor_tables_of_their_own(tmp_path):
    """The pool's threads leave the process's descriptor table (its lock bounded the reader at ~0.8 M files/s on 16 CPUs) for an empty
    private one: they keep no reference to what the process has open -- the peer of a socket closed here sees the end at once -- and the
    files they read still arrive whole."""
    import socket
    a, b = socket.socketpair()                               # open BEFORE the pool starts: a copied table would keep `a` alive
    r = np.random.default_rng(5)
    paths, payloads = [], []
    for i in range(64):
        raw = r.integers(-2 ** 15, 2 ** 15 - 1, 500 + i).astype("<i2").tobytes()
        p = os.path.join(tmp_path, f"p{i:02d}.wav")
        with open(p, "wb") as f:
            f.write(_wav(raw))
        paths.append(p); payloads.append(raw)
    rd = files.WavBatchReader(max_clips=64, max_raw_bytes=1 << 20, threads=6, slots=2, host_only=True)
    descs, status = rd.read(paths, 0)                        # every pool thread has run by now
    assert (np.asarray(status) == 1).all()
    stage = rd.staging(0)
    for d, raw in zip(descs, payloads):
        assert bytes(stage[int(d["byte_offset"]):int(d["byte_offset"]) + len(raw)]) == raw
    tables = [set(os.listdir(f"/proc/self/task/{t}/fd")) for t in os.listdir("/proc/self/task")]
    assert sum(t <= {"0", "1", "2"} for t in tables) == 5, tables      # the five pool threads (the caller is the sixth worker)
    a.close()
    b.settimeout(5.0)
    assert b.recv(1) == b""                                  # EOF: nobody else holds the other end
    b.close()
    before = set(os.listdir("/proc/self/fd"))
    rd.read(paths, 1)
    assert set(os.listdir("/proc/self/fd")) == before        # nothing the pool opened stays behind in the process's table
    rd.close()
