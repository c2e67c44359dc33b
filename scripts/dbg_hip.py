import ctypes as C, torch, os
print("torch", torch.__version__, torch.cuda.is_available(), torch.cuda.device_count())
torch.cuda.set_device(0)
import wakeword_jupyterlab_amd as pkg
from wakeword_jupyterlab_amd import _native as nat
for line in open('/proc/self/maps'):
    if 'amdhip64' in line and 'r-xp' in line: print(line.strip())
n = C.c_int(-1)
for path in [None]:
    h = C.CDLL("libamdhip64.so.7")
    rc = h.hipGetDeviceCount(C.byref(n)); print("hipGetDeviceCount via soname:", rc, n.value)
print("ww_init:", nat.lib.ww_init(), nat.lib.ww_last_error())
