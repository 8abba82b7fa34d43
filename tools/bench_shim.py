"""PCIe-inclusive rate of the FFTW-named host-pointer boundary (include/fftw3.h): fftwf_execute on a pinned 4K RGB frame =
H2D + two axis passes + D2H, synchronous.  python tools/bench_shim.py"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dspfun_amd import _lib
lib = C.CDLL(_lib.LIB_PATH)
ip = C.POINTER(C.c_int)
lib.fftwf_alloc_real.restype = C.c_void_p; lib.fftwf_alloc_real.argtypes = [C.c_size_t]
lib.fftwf_plan_many_r2r.restype = C.c_void_p
lib.fftwf_plan_many_r2r.argtypes = [C.c_int, ip, C.c_int, C.c_void_p, ip, C.c_int, C.c_int, C.c_void_p, ip, C.c_int, C.c_int, ip, C.c_uint]
lib.fftwf_execute.argtypes = [C.c_void_p]
h, w, c = 2160, 3840, 3
n = h * w * c
p = lib.fftwf_alloc_real(n)
a = np.ctypeslib.as_array((C.c_float * n).from_address(p)); a[:] = np.random.rand(n).astype(np.float32)
ia = lambda v: (C.c_int * len(v))(*v)
fwd = lib.fftwf_plan_many_r2r(2, ia([h, w]), c, p, None, c, 1, p, None, c, 1, ia([5, 5]), 1 << 6)
inv = lib.fftwf_plan_many_r2r(2, ia([h, w]), c, p, None, c, 1, p, None, c, 1, ia([4, 4]), 1 << 6)
for _ in range(3): lib.fftwf_execute(fwd); lib.fftwf_execute(inv); a *= np.float32(1.0 / (4.0 * w * h))
t0 = time.perf_counter()
R = 10
for _ in range(R): lib.fftwf_execute(fwd)
dt = (time.perf_counter() - t0) / R
# scan's per-frame plan (scan/scan.c:359,447): out of place, dense output (onembed == NULL) -> input up, output down: two transfers
q = lib.fftwf_alloc_real(n)
oop = lib.fftwf_plan_many_r2r(2, ia([h, w]), c, p, None, c, 1, q, None, c, 1, ia([4, 4]), 0)
for _ in range(2): lib.fftwf_execute(oop)
t0 = time.perf_counter()
for _ in range(R): lib.fftwf_execute(oop)
dt2 = (time.perf_counter() - t0) / R
print(json.dumps({"what": "fftwf_execute, 3840x2160x3 f32, OUT OF PLACE with a dense output (scan.c:359 P5)", "ms": round(dt2 * 1e3, 3),
                  "transfers_per_execute": 2, "host_GBps_each_way": round(n * 4 / dt2 / 1e9, 1),
                  "pcie_floor_ms_at_63GBps": round(2 * n * 4 / 63e9 * 1e3, 2)}))
print(json.dumps({"what": "fftwf_execute, 3840x2160x3 f32, pinned host buffer (fftwf_alloc_real), one direction", "ms": round(dt * 1e3, 3),
                  "Mpixels_per_s": round(h * w / dt / 1e6), "host_GBps_each_way": round(n * 4 / dt / 1e9 * 2 / 2, 1),
                  "note": "H2D + 2 kernels (~0.09 ms) + D2H, synchronous on return"}))
