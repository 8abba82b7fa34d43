"""CPU: host/scan_orders.c (the harness-side scan library) against the oracle restatements and the reference's
known answers.  Integer work: bit-exact."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import oracle_lib as ol
import scan_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
METHODS = ["horizontal", "vertical", "zigzag", "row", "column", "diagonal", "mirror", "box", "ibox", "radial", "iradial"]


@pytest.fixture(scope="module")
def so():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), "libscanorders.so"])
    lib = C.CDLL(os.path.join(ROOT, "host", "libscanorders.so"))
    lib.scan_order_limit.restype = C.c_size_t
    lib.scan_order_limit.argtypes = [C.c_int, C.c_size_t, C.c_size_t]
    lib.scan_order_max_interval.restype = C.c_size_t
    lib.scan_order_max_interval.argtypes = [C.c_int, C.c_size_t, C.c_size_t]
    lib.scan_order_coords.restype = C.c_size_t
    lib.scan_order_coords.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    lib.scan_order_find_prefix.argtypes = [C.c_char_p]
    lib.scan_order_serialize_coordinate.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
    lib.scan_order_serialize_index.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
    return lib


def product_orders(so, method, w, h):
    m = METHODS.index(method)
    buf = np.zeros((so.scan_order_max_interval(m, w, h) + 1, 2), dtype=np.uint64)
    out = []
    for i in range(so.scan_order_limit(m, w, h)):
        n = so.scan_order_coords(m, w, h, i, buf.ctypes.data)
        out.append([(int(buf[j, 0]), int(buf[j, 1])) for j in range(n)])
    return out


def oracle_orders(method, w, h):
    L = ol.lib()
    if method in ("mirror", "box", "ibox", "radial", "iradial"):
        return scan_ref.orders(method, w, h)
    if method in ("horizontal", "vertical", "zigzag"):
        fn = getattr(L, "oracle_scan_" + method)
        fn.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
        yx = (C.c_size_t * 2)()
        out = []
        for i in range(w * h):
            fn(w, h, i, yx)
            out.append([(yx[0], yx[1])])
        return out
    fn = getattr(L, "oracle_scan_" + method)
    fn.restype = C.c_size_t
    fn.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    limit = {"row": h, "column": w, "diagonal": w + h - 1}[method]
    buf = np.zeros((w + h + 1, 2), dtype=np.uint64)
    out = []
    for i in range(limit):
        n = fn(w, h, i, buf.ctypes.data)
        out.append([(int(buf[j, 0]), int(buf[j, 1])) for j in range(n)])
    return out


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("w,h", [(8, 8), (16, 9), (9, 16), (5, 1), (1, 5), (7, 3)])
def test_scan_orders_match_oracle(so, method, w, h):
    assert product_orders(so, method, w, h) == oracle_orders(method, w, h)


def test_box_quirks_recorded_by_the_survey(so):
    # `box` emits 84 out-of-range x on 9x16 and 84 duplicate coordinates on 16x9
    o = [c for idx in product_orders(so, "box", 9, 16) for c in idx]
    assert sum(1 for (y, x) in o if x >= 9) == 84
    o = [c for idx in product_orders(so, "box", 16, 9) for c in idx]
    assert len(o) - len(set(o)) == 84
    # `ibox` emits its corner twice per index (scan_methods.c:135-144)
    o = product_orders(so, "ibox", 6, 4)
    assert all(idx.count((i, i)) == 2 for i, idx in enumerate(o))


def test_prefix_lookup(so):
    assert so.scan_order_find_prefix(b"z") == METHODS.index("zigzag")
    assert so.scan_order_find_prefix(b"r") == METHODS.index("row")          # shortest name with the prefix (scan_methods.c:581-591)
    assert so.scan_order_find_prefix(b"rad") == METHODS.index("radial")
    assert so.scan_order_find_prefix(b"nope") == -1


def test_serialisers_reproduce_readme_listings(so, scan_golden):
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    m = METHODS.index("diagonal")
    for fn, key in ((so.scan_order_serialize_index, "diagonal_8x8_index"), (so.scan_order_serialize_coordinate, "diagonal_8x8_coordinate")):
        with tempfile.NamedTemporaryFile(suffix=".txt", delete=False) as tf:
            path = tf.name
        f = libc.fopen(path.encode(), b"w")
        assert fn(m, 8, 8, f) == 0
        libc.fclose(f)
        lines = [ln.rstrip() for ln in open(path).read().splitlines()]
        os.unlink(path)
        assert lines == [ln.rstrip() for ln in scan_golden[key]]


@pytest.mark.parametrize("method", ["horizontal", "vertical", "zigzag", "row", "column", "diagonal", "mirror", "radial", "iradial"])
def test_permutation_methods_cover_every_pixel_once(so, method):
    w, h = 12, 7
    flat = [c for idx in product_orders(so, method, w, h) for c in idx]
    assert sorted(flat) == [(y, x) for y in range(h) for x in range(w)]


class _OrderList(C.Structure):
    _fields_ = [("limit", C.c_size_t), ("max_interval", C.c_size_t), ("total", C.c_size_t), ("offset", C.POINTER(C.c_size_t)), ("yx", C.POINTER(C.c_size_t * 2))]


class _Precomputed(C.Structure):          # scan/scan_precomputed.h:12-16
    _fields_ = [("limit", C.c_size_t), ("intervals", C.POINTER(C.c_size_t)), ("scans", C.POINTER(C.POINTER(C.c_size_t * 2)))]


@pytest.mark.parametrize("method,w,h", [("diagonal", 8, 8), ("box", 16, 9), ("ibox", 9, 16), ("mirror", 7, 5), ("zigzag", 6, 4), ("radial", 10, 6)])
@pytest.mark.parametrize("fmt", ["coordinate", "index"])
def test_file_method_reader_roundtrips_and_matches_the_reference_deserialiser(so, method, w, h, fmt):
    """`file` (scan_methods.c:393-410): host/scan_orders.c scan_order_read_file on both plaintext serialisations, against the
    generator that wrote the file and against the REFERENCE's own scan_precomputed_unserialize (oracle/_ref)."""
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    so.scan_order_read_file.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(_OrderList)]
    so.scan_order_list_free.argtypes = [C.POINTER(_OrderList)]
    m = METHODS.index(method)
    orders = product_orders(so, method, w, h)
    if method == "box":
        if fmt == "index":
            pytest.skip("an index grid holds one scan index per pixel: box's shared pixels do not survive it")
        w_file = max(x for cs in orders for (_, x) in cs) + 1       # box on tall frames runs past the width: the file is wider
    else:
        w_file = w
    with tempfile.NamedTemporaryFile(suffix=".txt", delete=False) as tf:
        path = tf.name
    f = libc.fopen(path.encode(), b"w")
    assert (so.scan_order_serialize_coordinate if fmt == "coordinate" else so.scan_order_serialize_index)(m, w, h, f) == 0
    libc.fclose(f)
    lst = _OrderList()
    f = libc.fopen(path.encode(), b"r")
    assert so.scan_order_read_file(f, max(w, w_file), h, C.byref(lst)) == 0
    libc.fclose(f)
    got = [[(lst.yx[k][0], lst.yx[k][1]) for k in range(lst.offset[i], lst.offset[i + 1])] for i in range(lst.limit)]
    if fmt == "coordinate":
        assert got == orders
    else:        # the grid stores, per pixel, the LAST index that emitted it, listed in raster order
        owner = {}
        for i, cs in enumerate(orders):
            for c in cs:
                owner[c] = i
        want = [[] for _ in range(len(orders))]
        for y in range(h):
            for x in range(w):
                want[owner[(y, x)]].append((y, x))
        assert got == want
    # the reference's deserialiser on the same file
    r = ol.ref()
    if r is not None:
        r.scan_precomputed_unserialize.restype = C.POINTER(_Precomputed)
        r.scan_precomputed_unserialize.argtypes = [C.c_void_p]
        f = libc.fopen(path.encode(), b"r")
        p = r.scan_precomputed_unserialize(f)
        libc.fclose(f)
        assert p and p.contents.limit == lst.limit
        refl = [[(p.contents.scans[i][j][0], p.contents.scans[i][j][1]) for j in range(p.contents.intervals[i])] for i in range(p.contents.limit)]
        assert refl == got
    so.scan_order_list_free(C.byref(lst))
    # out-of-range coordinates are rejected like init_file does
    f = libc.fopen(path.encode(), b"r")
    assert so.scan_order_read_file(f, max(1, w - 1), h, C.byref(lst)) != 0
    libc.fclose(f)
    os.unlink(path)


def test_box_ibox_max_interval_is_limit_sum(so):
    """scan_methods.c:23,496,502: box and ibox report limit_sum = w + h - 1, although ibox's index 0 emits its corner twice (w + h
    coordinates) -- the tool's buffer has one entry to spare (scan.c:346).  The value feeds scan.c:349-350's use_fftw selector."""
    for (w, h) in ((16, 9), (9, 16), (8, 8), (1920, 1080)):
        for name in ("box", "ibox"):
            assert so.scan_order_max_interval(METHODS.index(name), w, h) == w + h - 1
    buf = np.zeros((16 + 9 + 1, 2), dtype=np.uint64)
    assert so.scan_order_coords(METHODS.index("ibox"), 16, 9, 0, buf.ctypes.data) == 16 + 9


def test_random_order_is_the_references_permutation(so):
    """scan_methods.c:210-228 init_random compiled as it lies (tests/golden/make_ref_fixtures.py) against host/scan_orders.c's
    scan_order_random: same libc rand() stream, same Fisher-Yates loop (ends above index 1), coordinates ctx[i] / w, ctx[i] % w."""
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_direct.npz"))
    so.scan_order_random.argtypes = [C.c_size_t, C.c_size_t, C.c_uint, C.POINTER(_OrderList)]
    so.scan_order_list_free.argtypes = [C.POINTER(_OrderList)]
    for i, (w, h, seed) in enumerate(fx["random_cases"]):
        w, h, seed = int(w), int(h), int(seed)
        lst = _OrderList()
        assert so.scan_order_random(w, h, seed, C.byref(lst)) == 0
        assert lst.limit == w * h and lst.max_interval == 1 and lst.total == w * h
        yx = np.array([[lst.yx[k][0], lst.yx[k][1]] for k in range(w * h)], dtype=np.uint64)
        perm = fx[f"random{i}_perm"]
        assert np.array_equal(yx[:, 0] * np.uint64(w) + yx[:, 1], perm)
        assert sorted(perm.tolist()) == list(range(w * h))
        so.scan_order_list_free(C.byref(lst))
