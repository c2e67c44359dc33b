"""Test-side WAV reading (numpy only): an independent reader for the tests that compare the native reader + K0 with oracle/decode_oracle.py.
Not product code: the package reads files with csrc/ww_files.cpp and decodes on the GPU."""
import numpy as np


def _parse_wav(data: bytes):
    """RIFF/WAVE header walk -> (format tag, channels, sample rate, bits, data offset, data length)."""
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, where = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], int.from_bytes(data[pos + 4:pos + 8], "little")
        if cid == b"fmt ":
            body = data[pos + 8:pos + 8 + size]
            tag, ch, sr = int.from_bytes(body[0:2], "little"), int.from_bytes(body[2:4], "little"), int.from_bytes(body[4:8], "little")
            bits = int.from_bytes(body[14:16], "little")
            if tag == 0xFFFE and len(body) >= 26:
                tag = int.from_bytes(body[24:26], "little")
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            if size in (0, 0xFFFFFFFF):                    # unfinalised / streaming header: to the end of the file (libsndfile's reading)
                where = (pos + 8, len(data) - pos - 8)
                break
            where = (pos + 8, min(size, len(data) - pos - 8))
        pos += 8 + size + (size & 1)
    if fmt is None or where is None:
        raise ValueError("missing fmt/data chunk")
    return fmt + where


def _read_wav(path: str):
    """Minimal RIFF/WAVE reader: 8/16/24/32-bit PCM and 32-bit float, any channel count -> (float32 [n, ch], sr)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], int.from_bytes(data[pos + 4:pos + 8], "little")
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr = int.from_bytes(body[0:2], "little"), int.from_bytes(body[2:4], "little"), int.from_bytes(body[4:8], "little")
            bits = int.from_bytes(body[14:16], "little")
            if tag == 0xFFFE and len(body) >= 26:          # WAVE_FORMAT_EXTENSIBLE: real tag in the GUID
                tag = int.from_bytes(body[24:26], "little")
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            if size in (0, 0xFFFFFFFF):
                pcm = data[pos + 8:]
                break
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError("missing fmt/data chunk")
    tag, ch, sr, bits = fmt
    if tag == 1 and bits == 8:
        x = (np.frombuffer(pcm, np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm[: len(pcm) // 2 * 2], "<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm[: len(pcm) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (np.where(v >= 1 << 23, v - (1 << 24), v)).astype(np.float32) / float(1 << 23)
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], "<i4").astype(np.float32) / float(1 << 31)
    elif tag == 3 and bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], "<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(pcm[: len(pcm) // 8 * 8], "<f8").astype(np.float32)
    else:
        raise ValueError(f"unsupported WAV encoding tag={tag} bits={bits}")
    x = x[: len(x) // ch * ch].reshape(-1, ch)
    return x, sr
