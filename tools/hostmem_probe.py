import time, mmap, ctypes, numpy as np, threading, os
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
N = 256 << 20
def t(f, what):
    t0 = time.perf_counter(); r = f(); print(f"{what:50s} {(time.perf_counter()-t0)*1e3:8.2f} ms", flush=True); return r
a = t(lambda: np.empty(N, np.uint8), "np.empty 256 MiB")
t(lambda: a.__setitem__(slice(None, None, 4096), 1), "touch every page (np.empty)")
b = t(lambda: np.zeros(N, np.uint8), "np.zeros 256 MiB")
t(lambda: b.__setitem__(slice(None, None, 4096), 1), "touch every page (np.zeros)")
libc = ctypes.CDLL("libc.so.6", use_errno=True)
libc.mmap.restype = ctypes.c_void_p
libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
def mm(huge, populate=False):
    flags = 0x22 | (0x8000 if populate else 0)   # MAP_PRIVATE|MAP_ANONYMOUS (|MAP_POPULATE)
    p = libc.mmap(None, N + (2 << 20), 3, flags, -1, 0)
    q = (p + (2 << 20) - 1) & ~((2 << 20) - 1)
    if huge: print("  madvise ->", libc.madvise(q, N, 14))
    return np.ctypeslib.as_array((ctypes.c_uint8 * N).from_address(q))
c = t(lambda: mm(True), "mmap + MADV_HUGEPAGE")
t(lambda: c.__setitem__(slice(None, None, 4096), 1), "touch every page (THP hint)")
d = t(lambda: mm(False), "mmap plain")
def par_touch(arr, nt=8):
    step = N // nt
    th = [threading.Thread(target=lambda i=i: arr[i*step:(i+1)*step:4096].fill(1)) for i in range(nt)]
    [x.start() for x in th]; [x.join() for x in th]
t(lambda: par_touch(d), "touch every page, 8 threads (plain)")
e = mm(True)
t(lambda: par_touch(e), "touch every page, 8 threads (THP hint)")
f = t(lambda: mm(False, True), "mmap MAP_POPULATE")
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from wavelets_amd import _lib as L
ctx = L.default_context()
h = t(lambda: L.host_empty((8192, 8192), ctx, np.float32), "host_empty (hipHostMalloc 256 MiB)")
hip = ctypes.CDLL("libamdhip64.so")
g = mm(True); par_touch(g)
t(lambda: print("  rc", hip.hipHostRegister(ctypes.c_void_p(g.ctypes.data), ctypes.c_size_t(N), 0)), "hipHostRegister of a faulted THP region")
g2 = mm(False); par_touch(g2)
t(lambda: print("  rc", hip.hipHostRegister(ctypes.c_void_p(g2.ctypes.data), ctypes.c_size_t(N), 0)), "hipHostRegister of a faulted 4K-page region")
