"""Does this host allow io_uring_setup? (the reader would batch open/read/close through it)"""
import ctypes, os
libc = ctypes.CDLL(None, use_errno=True)
buf = ctypes.create_string_buffer(256)
r = libc.syscall(425, 8, buf)
print("io_uring_setup ->", r, os.strerror(ctypes.get_errno()) if r < 0 else "ok", "| kernel", os.uname().release)
