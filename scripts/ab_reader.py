"""A/B of the batch WAV reader's host half on the GPU box: files/s of ww_read_wav_batch_host for several builds of the library, interleaved.
usage: PYTHONPATH=. python scripts/ab_reader.py libA.so libB.so ..."""
import ctypes as C, os, shutil, struct, sys, tempfile, time
import numpy as np

libs = sys.argv[1:]
d = tempfile.mkdtemp(prefix="ww_ab_reader_")
raw = np.random.default_rng(0).integers(-2 ** 15, 2 ** 15 - 1, 16000).astype("<i2").tobytes()
hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data" + struct.pack("<I", len(raw))
paths = []
for i in range(4096):
    p = os.path.join(d, f"{i}.wav")
    with open(p, "wb") as f:
        f.write(hdr + raw)
    paths.append(os.fsencode(p))
arr = (C.c_char_p * 4096)(*paths)
status = (C.c_int8 * 4096)()
hs = []
for path in libs:
    h = C.CDLL(path)
    rd = C.c_void_p()
    assert h.ww_wav_reader_create(C.c_int32(16), C.c_int32(2), C.c_int64(4096), C.c_int64(4096 * 32064), C.c_int32(1), C.byref(rd)) == 0
    hs.append((path, h, rd))
res = {p: [] for p in libs}
for rnd in range(8):
    for path, h, rd in hs:
        descs, need = C.c_void_p(), C.c_int64()
        t = time.perf_counter()
        for k in range(4):
            assert h.ww_read_wav_batch_host(rd, arr, C.c_int64(4096), C.c_int32(k & 1), C.byref(descs), status, C.byref(need)) == 0
        res[path].append(4 * 4096 / (time.perf_counter() - t))
for p in libs:
    v = np.array(res[p][1:])
    print("%-40s median %.0f files/s  max %.0f" % (p.split("/")[-1], np.median(v), v.max()))
shutil.rmtree(d)
