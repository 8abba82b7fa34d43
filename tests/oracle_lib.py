"""ctypes bindings to oracle/liboracle.so (and oracle/_ref) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under dspfun_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REDFT01, REDFT10 = 4, 5

_lib = None
_ref = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        ip = C.POINTER(C.c_int)
        _lib.oracle_r2r_many_f64.argtypes = [C.c_int, ip, C.c_int, C.c_void_p, ip, C.c_int, C.c_int,
                                             C.c_void_p, ip, C.c_int, C.c_int, ip]
        for suf, _ in (("f32", np.float32), ("f64", np.float64)):
            f = getattr(_lib, "cpu_port_r2r_many_" + suf)
            f.argtypes = [C.c_int, ip, C.c_int, C.c_void_p, ip, C.c_int, C.c_int,
                          C.c_void_p, ip, C.c_int, C.c_int, ip, C.c_int]
        _lib.oracle_zigzag_fnv.restype = C.c_uint64
        _lib.oracle_zigzag_fnv.argtypes = [C.c_size_t, C.c_size_t]
        _lib.oracle_zigzag_order.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p]
        _lib.oracle_fnv1a64_u64.restype = C.c_uint64
        _lib.oracle_fnv1a64_u64.argtypes = [C.c_void_p, C.c_size_t]
        _lib.oracle_scan_zigzag.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        _lib.oracle_scan_frame_f64.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.oracle_spec_normalise_f64.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.oracle_ispec_denormalise_f64.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.oracle_scan_normalise_f64.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.oracle_motion_uniform_f64.argtypes = [C.c_void_p] + [C.c_int] * 6
        _lib.oracle_motion_store_u8_f64.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 8
        _lib.oracle_zoom_basis_f64.restype = C.c_size_t
        _lib.oracle_zoom_basis_f64.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_size_t, C.c_size_t]
        _lib.oracle_zoom_product_f64.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                 C.c_void_p, C.c_int, C.c_int]
        _lib.cpu_port_max_threads.restype = C.c_int
    return _lib


def ref():
    """The reference's own compilable TUs (oracle/_ref/libdspfun_ref.so), or None."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libdspfun_ref.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
    return _ref


def _ia(v):
    if v is None:
        return None
    return (C.c_int * len(v))(*[int(x) for x in v])


def _span(n, embed, stride, dist, howmany):
    e = list(embed) if embed is not None else list(n)
    last = 0
    idx = 0
    for a in range(len(n)):
        idx = idx * e[a] + (n[a] - 1)
    last = idx * stride + (howmany - 1) * dist
    return last + 1


def r2r_many(x, n, kinds, howmany=1, inembed=None, istride=1, idist=0, onembed=None, ostride=1, odist=0,
             out=None, impl="direct", threads=1):
    """plan_many_r2r + execute on a flat array; returns the flat output array (same dtype as x
    for impl='port', float64 for impl='direct')."""
    n = list(n)
    kinds = list(kinds)
    L = lib()
    if impl == "direct":
        xin = np.ascontiguousarray(x, dtype=np.float64).ravel()
        need = _span(n, onembed, ostride, odist, howmany)
        if out is None:
            o = xin.copy() if need <= xin.size else np.zeros(need)
        else:
            o = np.ascontiguousarray(out, dtype=np.float64).ravel().copy()
        assert o.size >= need
        rc = L.oracle_r2r_many_f64(len(n), _ia(n), howmany, xin.ctypes.data, _ia(inembed), istride, idist,
                                   o.ctypes.data, _ia(onembed), ostride, odist, _ia(kinds))
        assert rc == 0
        return o
    dt = np.dtype(x.dtype)
    assert dt in (np.float32, np.float64)
    xin = np.ascontiguousarray(x).ravel()
    need = _span(n, onembed, ostride, odist, howmany)
    if out is None:
        o = xin.copy() if need <= xin.size else np.zeros(need, dtype=dt)
    else:
        o = np.ascontiguousarray(out, dtype=dt).ravel().copy()
    f = getattr(L, "cpu_port_r2r_many_" + ("f32" if dt == np.float32 else "f64"))
    rc = f(len(n), _ia(n), howmany, xin.ctypes.data, _ia(inembed), istride, idist,
           o.ctypes.data, _ia(onembed), ostride, odist, _ia(kinds), threads)
    assert rc == 0
    return o


def dct2d_interleaved(img, kind, impl="direct", threads=1):
    """img: (h, w, c) array; the spec/scan/zoom plan shape (howmany=c, stride=c, dist=1)."""
    h, w, c = img.shape
    o = r2r_many(img, [h, w], [kind, kind], howmany=c, istride=c, idist=1, ostride=c, odist=1, impl=impl, threads=threads)
    return o.reshape(h, w, c)


def zigzag_order(w, h):
    out = np.empty(w * h, dtype=np.uint64)
    lib().oracle_zigzag_order(w, h, out.ctypes.data)
    return out


def splitmix64_stream(seed, n):
    """state advances once per sample; returns uint64[n]."""
    mask = (1 << 64) - 1
    out = np.empty(n, dtype=np.uint64)
    # vectorised: state_i = seed + (i+1)*golden
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & mask) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    out[:] = z
    return out


def synth_f32(seed, n):
    """f32 in [0,1): (u >> 40) * 2^-24"""
    u = splitmix64_stream(seed, n)
    return ((u >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def synth_u8(seed, n):
    u = splitmix64_stream(seed, n)
    return (u >> np.uint64(56)).astype(np.uint8)
