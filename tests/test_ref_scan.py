"""CPU: every scan method pinned to the REFERENCE'S OWN code (the device generators were only tested against the
product's host library).  tests/golden/ref_scan.npz comes from scan/scan_methods.c:16-184,203-331 and scan/scan_precomputed.c compiled
as they lie by tests/golden/make_ref_fixtures.py (everything but the libavutil-based evalxy / evali).  Checked here: the host library
(host/scan_orders.c) and, through the CPU emulation of the device kernels, dspfft_scan_owner_index / dspfft_scan_coords /
dspfft_scan_magnitude_index.  The GPU run of the same kernels is tests/test_scan_methods_gpu.py (same assertions against the host library)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from emul_lib import emul

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
METHODS = ["horizontal", "vertical", "zigzag", "row", "column", "diagonal", "mirror", "box", "ibox", "radial", "iradial"]
NONE = 0xFFFFFFFF


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_scan.npz"))


@pytest.fixture(scope="module")
def so():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), "libscanorders.so"])
    lib = C.CDLL(os.path.join(ROOT, "host", "libscanorders.so"))
    for f in (lib.scan_order_limit, lib.scan_order_max_interval):
        f.restype = C.c_size_t
        f.argtypes = [C.c_int, C.c_size_t, C.c_size_t]
    lib.scan_order_coords.restype = C.c_size_t
    lib.scan_order_coords.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    return lib


def fnv_words(words):
    """the fixture's order-sensitive checksum: sum_k word_k * (2k + 1) mod 2^64"""
    wv = np.asarray(words, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return int((wv * (np.arange(len(wv), dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))


def host_lists(so, m, w, h):
    lim = so.scan_order_limit(m, w, h)
    buf = np.zeros((so.scan_order_max_interval(m, w, h) + 2, 2), dtype=np.uint64)
    counts, flat = [], []
    for i in range(lim):
        n = so.scan_order_coords(m, w, h, i, buf.ctypes.data)
        counts.append(n)
        flat.append(buf[:n].copy())
    return lim, counts, (np.concatenate(flat) if flat else np.zeros((0, 2), dtype=np.uint64))


@pytest.mark.parametrize("name", METHODS[:9])
def test_host_generators_equal_the_reference(fx, so, name):
    m = METHODS.index(name)
    big = [tuple(int(v) for v in s) for s in fx["big_sizes"]]
    if m < 3:
        big = [(640, 480)]       # one index per pixel: 2 M calls at full HD (zigzag at 1080p / 8K is pinned by the survey's hashes, tests/test_oracle.py)
    for (w, h) in [tuple(int(v) for v in s) for s in fx["small_sizes"]] + big:
        key = f"{name}_{w}x{h}"
        rlim, rmi, rtotal = (int(v) for v in fx[key + "_meta"])
        lim, counts, flat = host_lists(so, m, w, h)
        assert lim == rlim and so.scan_order_max_interval(m, w, h) == rmi and len(flat) == rtotal, key
        if key + "_yx" in fx:
            assert np.array_equal(np.array(counts, dtype=np.uint32), fx[key + "_counts"]) and np.array_equal(flat.astype(np.uint32), fx[key + "_yx"]), key
        else:
            assert [fnv_words(counts), fnv_words((flat[:, 0] << np.uint64(32)) | flat[:, 1])] == [int(v) for v in fx[key + "_fnv"]], key


@pytest.mark.parametrize("name", ["radial", "iradial"])
def test_host_radial_orders_equal_the_reference(fx, so, name):
    m = METHODS.index(name)
    for (w, h) in [tuple(int(v) for v in s) for s in fx["small_sizes"]] + [(640, 480)]:
        key = f"{name}_{w}x{h}"
        lim, counts, flat = host_lists(so, m, w, h)
        assert lim == int(fx[key + "_limit"][0]), key
        lin = (flat[:, 0] * np.uint64(w) + flat[:, 1]).astype(np.uint32)
        idx = np.zeros(w * h, dtype=np.uint32)
        idx[lin] = np.repeat(np.arange(lim, dtype=np.uint32), counts)
        if key + "_index" in fx:
            assert np.array_equal(idx, fx[key + "_index"]) and np.array_equal(lin, fx[key + "_order"]), key
        else:
            assert [fnv_words(idx), fnv_words(lin)] == [int(v) for v in fx[key + "_fnv"]], key


def _dev(L, fn, n, *args):
    buf = np.zeros(n, dtype=np.uint32)
    assert fn(buf.ctypes.data, *args, None) == 0, L.dspfft_last_error()
    return buf


def test_device_kernels_equal_the_reference_through_the_emulation(fx):
    L = emul()
    for name in METHODS:
        m = METHODS.index(name)
        for (w, h) in [(16, 9), (9, 16), (33, 20), (8, 8)]:
            key = f"{name}_{w}x{h}"
            if name in ("radial", "iradial"):
                got = _dev(L, L.dspfft_scan_owner_index, w * h, m, w, h)
                assert np.array_equal(got, fx[key + "_index"]), key
                assert L.dspfft_scan_limit(m, w, h) == int(fx[key + "_limit"][0])
                continue
            rlim, rmi, _ = (int(v) for v in fx[key + "_meta"])
            counts, yx = fx[key + "_counts"], fx[key + "_yx"].astype(np.int64)
            assert L.dspfft_scan_limit(m, w, h) == rlim and L.dspfft_scan_max_interval(m, w, h) == rmi, key
            slots = L.dspfft_scan_coord_slots(m, w, h)
            got = _dev(L, L.dspfft_scan_coords, rlim * slots, m, w, h, 0, rlim).reshape(rlim, slots)
            o = 0
            for i, n in enumerate(counts):
                lin = yx[o:o + n, 0] * w + yx[o:o + n, 1]
                exp = np.where(lin < w * h, lin, NONE)           # box on tall frames: first leg past the end of the image (scan_methods.c:122-133)
                assert np.array_equal(got[i, :n].astype(np.int64), exp) and np.all(got[i, n:] == NONE), (key, i)
                o += n
            if name != "box":                                    # one owner per pixel (ibox's doubled corner has the same owner)
                own = np.zeros(w * h, dtype=np.uint32)
                own[yx[:, 0] * w + yx[:, 1]] = np.repeat(np.arange(rlim, dtype=np.uint32), counts)
                assert np.array_equal(_dev(L, L.dspfft_scan_owner_index, w * h, m, w, h), own), key


@pytest.mark.parametrize("name", METHODS[:3])
def test_one_index_per_pixel_orders_at_baseline_frame_sizes(fx, name):
    """horizontal, vertical and zigzag at 3840 x 2160 (BASELINE config 2's frame) through the device kernels' per-element function (emulation),
    hashed as the reference's compiled scan_methods.c hashed them (ref_scan.npz *_fnv1a); 7680 x 4320: tests/test_gpu_parity.py, on the GPU"""
    import oracle_lib as ol
    L = emul()
    m = METHODS.index(name)
    for (w, h) in [(3840, 2160), (256, 256)]:
        lin = _dev(L, L.dspfft_scan_coords, w * h, m, w, h, 0, w * h).astype(np.uint64)
        assert "%016x" % ol.lib().oracle_fnv1a64_u64(lin.ctypes.data, lin.size) == "%016x" % int(fx[f"{name}_{w}x{h}_fnv1a"][0]), (name, w, h)


@pytest.mark.parametrize("ci", range(4))
def test_magnitude_against_the_references_init_magnitude(fx, ci):
    """scan_methods.c:240-285 (qsort of the F/D-precision keys, then the grouping loop).  Distinct keys: the whole order is defined and
    must match.  Quantised keys: qsort leaves ties in an unspecified order, so the GROUPS (index -> set of pixels) are compared."""
    L = emul()
    w, h, q = (int(v) for v in fx[f"magnitude{ci}_shape"])
    coeffs = np.ascontiguousarray(fx[f"magnitude{ci}_coeffs"])
    idx = np.zeros(w * h, dtype=np.uint32)
    work = np.zeros(L.dspfft_scan_magnitude_work_bytes(w, h), dtype=np.uint8)
    lim = C.c_uint32()
    assert L.dspfft_scan_magnitude_index(idx.ctypes.data, coeffs.ctypes.data, w, h, 3, float(q), work.ctypes.data, work.size, C.byref(lim), None) == 0
    sizes, order = fx[f"magnitude{ci}_group_sizes"], fx[f"magnitude{ci}_order"]
    ref = np.zeros(w * h, dtype=np.uint32)
    ref[order] = np.repeat(np.arange(len(sizes), dtype=np.uint32), sizes)
    assert lim.value == len(sizes)
    assert np.array_equal(idx, ref)
