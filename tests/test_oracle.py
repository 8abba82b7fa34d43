"""CPU: pins the oracle (oracle/*.c) to the committed fixtures.  No GPU, no product code."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

import oracle_lib as ol

R10, R01 = ol.REDFT10, ol.REDFT01


@pytest.mark.parametrize("i", range(8))
@pytest.mark.parametrize("kind,name", [(R10, "redft10"), (R01, "redft01")])
def test_direct_oracle_2d_vs_golden(golden, i, kind, name):
    x = golden[f"img{i}_in"].astype(np.float64)
    ref = golden[f"img{i}_{name}"]
    got = ol.dct2d_interleaved(x, kind)
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("N", (2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 17, 27, 30, 31, 45, 60, 64, 97, 135, 270, 540))
def test_port_1d_vs_golden(golden, N):
    x = golden[f"vec{N}_in"]
    for kind, name in ((R10, "redft10"), (R01, "redft01")):
        ref = golden[f"vec{N}_{name}"]
        d = ol.r2r_many(x.astype(np.float64), [N], [kind])
        p64 = ol.r2r_many(x.astype(np.float64), [N], [kind], impl="port")
        p32 = ol.r2r_many(x, [N], [kind], impl="port")
        scale = max(1.0, np.abs(ref).max())
        assert np.abs(d - ref).max() <= 1e-12 * scale
        assert np.abs(p64 - ref).max() <= 1e-11 * scale
        assert np.abs(p32 - ref).max() <= 2e-6 * scale


def test_volume_embedded_vs_golden(golden):
    d, h, w, md, mh, mw = [int(v) for v in golden["vol_dims"]]
    buf = golden["vol_in"]
    for kind, name in ((R10, "redft10"), (R01, "redft01")):
        for impl in ("direct", "port"):
            o = ol.r2r_many(buf, [d, h, w], [kind] * 3, inembed=[md, mh, mw], onembed=[md, mh, mw], impl=impl)
            ref = golden[f"vol_{name}"]
            assert np.abs(o.reshape(md, mh, mw) - ref).max() <= 1e-9 * np.abs(ref).max()


def test_roundtrip_gain():
    # REDFT01(REDFT10(x)) = 2N x per axis (spec/spec.c:64 comment, scan/scan.c:296-298, motion.c:567)
    x = ol.synth_f32(1, 12 * 10 * 3).astype(np.float64).reshape(12, 10, 3)
    f = ol.dct2d_interleaved(x, R10)
    b = ol.dct2d_interleaved(f, R01)
    assert np.abs(b / (4 * 12 * 10) - x).max() < 1e-13


def test_spec_ispec_identity():
    # spec.c:63-78 then ispec.c:153-167 is the identity
    h, w, d = 9, 14, 3
    x = ol.synth_f32(2, h * w * d).astype(np.float64).reshape(h, w, d)
    f = np.ascontiguousarray(ol.dct2d_interleaved(x, R10))
    ol.lib().oracle_spec_normalise_f64(f.ctypes.data, w, h, d)
    assert np.abs(f).max() <= 1.0 + 1e-12 and abs(f[0, 0, 0] - x[:, :, 0].mean()) < 1e-12
    ol.lib().oracle_ispec_denormalise_f64(f.ctypes.data, w, h, d)
    y = ol.dct2d_interleaved(f, R01)
    assert np.abs(y - x).max() < 1e-13


def test_zigzag_hashes_match_reference(scan_golden):
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_scan.npz"))
    want_by_size = dict(scan_golden["zigzag_fnv"])
    for (w, h) in fx["fnv1a_sizes"]:        # the reference's scan_methods.c compiled without a stand-in header (make_ref_fixtures.py), 4K and 8K among them
        v = "%016x" % int(fx[f"zigzag_{w}x{h}_fnv1a"][0])
        assert want_by_size.setdefault(f"{w}x{h}", v) == v, (w, h)
    for key, want in want_by_size.items():
        w, h = [int(v) for v in key.split("x")]
        got = "%016x" % ol.lib().oracle_zigzag_fnv(w, h)
        assert got == want, key


@pytest.mark.parametrize("w,h", [(8, 8), (6, 4), (4, 6), (16, 9), (9, 16), (1, 5), (5, 1), (1, 1), (2, 2), (64, 48)])
def test_zigzag_is_permutation(w, h):
    o = ol.zigzag_order(w, h)
    assert sorted(o.tolist()) == list(range(w * h))


class _Method(C.Structure):
    # scan/scan_methods.h:12-24
    _fields_ = [("name", C.c_char_p), ("scan", C.c_void_p), ("limit", C.c_void_p), ("interval", C.c_void_p),
                ("max_interval", C.c_void_p), ("init_args", C.c_char_p), ("init", C.c_void_p), ("destroy", C.c_void_p)]


def _fp(name):
    return C.cast(getattr(ol.lib(), name), C.c_void_p)


def test_diagonal_known_answers_through_reference_serialiser(scan_golden):
    """scan/README.md:121-150: the restated `diagonal` generator, rendered by the reference's own
    scan_context.c/scan_precomputed.c (oracle/_ref), reproduces both listings byte for byte."""
    ref = ol.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    m = _Method(b"diagonal", _fp("oracle_method_diagonal"), _fp("oracle_limit_sum"), _fp("oracle_interval_diagonal"),
                _fp("oracle_limit_min"), None, None, None)
    ref.scan_init.restype = C.c_void_p
    ref.scan_init.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_char_p]
    ref.scan_serialize.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    ref.scan_serialize.restype = C.c_bool
    ref.scan_destroy.argtypes = [C.c_void_p]
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    ref.scan_serialization_val.restype = C.c_int
    ref.scan_serialization_val.argtypes = [C.c_char_p]
    for fmt, key in (("index", "diagonal_8x8_index"), ("coordinate", "diagonal_8x8_coordinate")):
        ctx = ref.scan_init(C.byref(m), 8, 8, 3, None, None)
        assert ctx
        with tempfile.NamedTemporaryFile(suffix=".txt", delete=False) as tf:
            path = tf.name
        f = libc.fopen(path.encode(), b"w")
        assert ref.scan_serialize(ctx, f, ref.scan_serialization_val(fmt.encode()))
        libc.fclose(f)
        ref.scan_destroy(ctx)
        lines = [ln.rstrip() for ln in open(path).read().splitlines()]
        os.unlink(path)
        assert lines == [ln.rstrip() for ln in scan_golden[key]], fmt


def test_scan_frames_sum_to_input():
    # scan/scan.c:377-383,421-459: DC broadcast + all masked inverse frames == the image
    w, h, c = 10, 6, 3
    x = ol.synth_f32(3, w * h * c).astype(np.float64).reshape(h, w, c)
    coeffs = np.ascontiguousarray(ol.dct2d_interleaved(x, R10))
    ol.lib().oracle_scan_normalise_f64(coeffs.ctypes.data, w, h, c)
    order = ol.zigzag_order(w, h)
    total = np.ascontiguousarray(np.broadcast_to(coeffs[0, 0], (h, w, c)).copy())
    step = 7
    for s in range(0, w * h, step):
        lin = np.ascontiguousarray(order[s:s + step])
        assert ol.lib().oracle_scan_frame_f64(w, h, c, coeffs.ctypes.data, lin.ctypes.data, lin.size, total.ctypes.data) == 0
    assert np.abs(total - x).max() < 1e-13


def test_motion_roundtrip_u8():
    # motion.c:617-647,748-776 with block == scaled: u8 in -> u8 out identical
    d, h, w = 4, 6, 8
    pix = ol.synth_u8(5, d * h * w)
    c = pix.astype(np.float64)
    c = ol.r2r_many(c, [d, h, w], [R10] * 3)
    ol.lib().oracle_motion_uniform_f64(c.ctypes.data, d, h, w, h, w, 1)
    mean = pix.astype(np.float64).mean()
    norm2 = 1.0 / (8 * d * h * w)
    assert abs(c[0] * norm2 - mean) < 1e-10          # u[0]*normalization^2 = mean
    ol.lib().oracle_motion_uniform_f64(c.ctypes.data, d, h, w, h, w, -1)
    c = ol.r2r_many(c, [d, h, w], [R01] * 3)
    out = np.zeros(d * h * w, dtype=np.uint8)
    ol.lib().oracle_motion_store_u8_f64(c.ctypes.data, out.ctypes.data, d, h, w, d, h, w, h, w)
    assert np.array_equal(out, pix)


def test_zoom_identity_and_integer_scale():
    # zoom.c:36-68,361-375: scale 1 is the identity; at integer scale the interpolated basis
    # reproduces the input samples at b = scale*q
    w, h = 12, 10
    x = ol.synth_f32(7, w * h * 3).astype(np.float64).reshape(h, w, 3)
    cf = np.ascontiguousarray(ol.dct2d_interleaved(x, R10))
    L = ol.lib()
    for scale in (1, 3):
        vw, vh = w * scale, h * scale
        cw = L.oracle_zoom_basis_f64(None, 0, scale, 1, 0.0, vw, w)
        ch = L.oracle_zoom_basis_f64(None, 0, scale, 1, 0.0, vh, h)
        xb = np.zeros(vw * (cw - 1)); yb = np.zeros(vh * (ch - 1))
        L.oracle_zoom_basis_f64(xb.ctypes.data, 0, scale, 1, 0.0, vw, w)
        L.oracle_zoom_basis_f64(yb.ctypes.data, 0, scale, 1, 0.0, vh, h)
        out = np.zeros((vh, vw, 3))
        L.oracle_zoom_product_f64(cf.ctypes.data, w, h, xb.ctypes.data, cw, yb.ctypes.data, ch, out.ctypes.data, vw, vh)
        assert np.abs(out[::scale, ::scale] - x).max() < 1e-12
