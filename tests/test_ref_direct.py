"""CPU: the oracle pinned to the REFERENCE'S OWN direct-sum code.

tests/golden/ref_direct.npz holds inputs and outputs of scan.c:20-41 (generate_basis_matrix + pruned_idct),
zoom.c:36-68,361-375 (generate_scaled_basis + the separable product) and applybasis.c:77-140,410-425 (basis functions,
partial sums) compiled as they lie by tests/golden/make_ref_fixtures.py (COEFF / INTERMEDIATE_PRECISION = L).  FFTW itself
was never run: what is pinned here is the reference's own statement of REDFT01 (and, through the 4wh roundtrip identity of
spec.c:64 / scan.c:296-298 and through applybasis' dct2, of REDFT10)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-12


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(HERE, "golden", "ref_direct.npz"))


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def test_dense_redft01_is_scans_pruned_idct_over_all_coefficients(fx):
    for i, (h, w, ch) in enumerate(fx["dense_shapes"]):
        c = fx[f"dense{i}_coeffs"]
        got = ol.dct2d_interleaved(c, ol.REDFT01)
        assert rel(got, fx[f"dense{i}_image"]) <= TOL, (h, w, ch)
        # the O(N log N) f64 port used as the checker at large sizes agrees too
        port = ol.dct2d_interleaved(c, ol.REDFT01, impl="port")
        assert rel(port, fx[f"dense{i}_image"]) <= TOL, (h, w, ch)


def test_redft10_through_the_4wh_roundtrip_identity(fx):
    """REDFT10(REDFT01_reference(c)) == 4wh c  (spec.c:64, scan.c:296-298): REDFT10 is pinned as 4wh x the inverse of the
    reference's own REDFT01."""
    for i, (h, w, ch) in enumerate(fx["dense_shapes"]):
        c, img = fx[f"dense{i}_coeffs"], fx[f"dense{i}_image"]
        for impl in ("direct", "port"):
            back = ol.dct2d_interleaved(img, ol.REDFT10, impl=impl)
            assert rel(back / (4.0 * w * h), c) <= TOL, (h, w, ch, impl)


def test_whole_operator_6x8(fx):
    op = fx["operator_6x8"]                        # column j = reference image of unit coefficient j
    h, w = 6, 8
    eye = np.eye(h * w)
    mine01 = np.stack([ol.dct2d_interleaved(eye[j].reshape(h, w, 1), ol.REDFT01).ravel() for j in range(h * w)], 1)
    assert np.abs(mine01 - op).max() <= TOL * np.abs(op).max()
    mine10 = np.stack([ol.dct2d_interleaved(eye[j].reshape(h, w, 1), ol.REDFT10).ravel() for j in range(h * w)], 1)
    assert np.abs(mine10 @ op - 4 * w * h * eye).max() <= 1e-11 * 4 * w * h


def test_1d_definitions_against_applybasis_dct2_dct3(fx):
    """REDFT10: Y_k = 2 sum_n x_n dct2(k, n, N); REDFT01: Y_k = 2 sum_n X_n dct3(k, n, N)  (applybasis.c:89-100; dct3(k, 0) = 1/2)"""
    for N in fx["basis_lens"]:
        N = int(N)
        x = ol.synth_f32(0xD5F1700 + N, N).astype(np.float64) * 2 - 1
        for kind, tab in ((ol.REDFT10, fx[f"dct2_{N}"]), (ol.REDFT01, fx[f"dct3_{N}"])):
            want = 2.0 * tab @ x
            for impl in ("direct", "port"):
                got = ol.r2r_many(x, [N], [kind], impl=impl)
                assert rel(got, want) <= TOL, (N, kind, impl)


def test_sparse_spectra_at_listed_frame_sizes(fx):
    """640x480, 1080p and 4K: the f64 port (the checker the GPU tests use at these sizes) against the reference's pruned_idct at
    sampled pixels."""
    for i, (h, w, ch, _, _) in enumerate(fx["sparse_shapes"]):
        c = np.zeros((h, w, ch))
        c[fx[f"sparse{i}_cy"], fx[f"sparse{i}_cx"]] = fx[f"sparse{i}_vals"]
        img = ol.dct2d_interleaved(c, ol.REDFT01, impl="port", threads=8)
        got = img[fx[f"sparse{i}_py"], fx[f"sparse{i}_px"]]
        assert rel(got, fx[f"sparse{i}_image_at"]) <= 1e-11, (h, w)
        if h <= 480:
            back = ol.dct2d_interleaved(img, ol.REDFT10, impl="port", threads=8) / (4.0 * w * h)
            assert np.abs(back - c).max() <= 1e-11


def zoom_oracle(c, typ, xn, xd, yn, yd, vx, vy, vw, vh):
    h, w, _ = c.shape
    L = ol.lib()
    cw = L.oracle_zoom_basis_f64(None, typ, xn, xd, vx, vw, w)
    ch = L.oracle_zoom_basis_f64(None, typ, yn, yd, vy, vh, h)
    xb = np.zeros(max(1, vw * (cw - 1))); yb = np.zeros(max(1, vh * (ch - 1)))
    L.oracle_zoom_basis_f64(xb.ctypes.data, typ, xn, xd, vx, vw, w)
    L.oracle_zoom_basis_f64(yb.ctypes.data, typ, yn, yd, vy, vh, h)
    out = np.zeros((vh, vw, 3))
    cf = np.ascontiguousarray(c)
    L.oracle_zoom_product_f64(cf.ctypes.data, w, h, xb.ctypes.data, cw, yb.ctypes.data, ch, out.ctypes.data, vw, vh)
    return out, (cw, ch)


def test_zoom_basis_and_product(fx):
    for i, case in enumerate(fx["zoom_cases"]):
        h, w, typ = int(case[0]), int(case[1]), int(case[2])
        xn, xd, yn, yd, vx, vy = (float(v) for v in case[3:])
        want = fx[f"zoom{i}_out"]
        vh, vw, _ = want.shape
        got, ncomp = zoom_oracle(fx[f"zoom{i}_coeffs"], typ, xn, xd, yn, yd, vx, vy, vw, vh)
        assert tuple(ncomp) == tuple(int(v) for v in fx[f"zoom{i}_ncomp"]), case
        assert rel(got, want) <= TOL, case


def test_zoom_at_scale_one_is_redft01_over_4wh(fx):
    """zoom.c:361-375 at scale 1, offset 0: a second reference-held statement of REDFT01"""
    c, want = fx["zoom0_coeffs"], fx["zoom0_out"]
    h, w, _ = c.shape
    assert rel(ol.dct2d_interleaved(c, ol.REDFT01) / (4.0 * w * h), want) <= TOL
    # native basis at integer scale == REDFT01 of the zero-padded spectrum
    c4, want4 = fx["zoom4_coeffs"], fx["zoom4_out"]
    vh, vw, _ = want4.shape
    pad = np.zeros((vh, vw, 3)); pad[:h, :w] = c4
    assert rel(ol.dct2d_interleaved(pad, ol.REDFT01) / (4.0 * w * h), want4) <= TOL


FUNCS = ["dft", "idft", "dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4", "wht", "dht"]


def test_applybasis_basis_functions(fx):
    L = ol.lib()
    L.oracle_applybasis_basis_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_ulonglong]
    for N in (4, 6, 8, 16):
        for ortho in (0, 1):
            tabs = fx[f"basis_all_{N}_{ortho}"]
            funcs = [f for f in range(12) if N != 6 or FUNCS[f] != "wht"]
            for t, f in zip(tabs, funcs):
                re = np.zeros((N, N)); im = np.zeros((N, N))
                L.oracle_applybasis_basis_f64(re.ctypes.data, im.ctypes.data, f, ortho, N, 0, N)
                assert np.abs(re + 1j * im - t).max() <= 1e-13, (FUNCS[f], N, ortho)


def test_applybasis_partial_sums_forward_inverse_offsets(fx):
    L = ol.lib()
    L.oracle_applybasis_partsums_ex_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 10 + [C.c_longlong, C.c_longlong, C.c_int]
    for i, case in enumerate(fx["parts_cases"]):
        f, ortho, inv, w, h, tw, th, pw, ph, ow, oh = (int(v) for v in case)
        want = fx[f"parts{i}_out"]
        Kh, Kw, Nh, Nw = want.shape[:4]
        pix = np.ascontiguousarray(fx[f"parts{i}_pix"])
        out = np.zeros((Kh, Kw, Nh, Nw, 3, 2))
        L.oracle_applybasis_partsums_ex_f64(out.ctypes.data, pix.ctypes.data, None, w, h, f, ortho, Kw, Kh, Nw, Nh, pw, ph, ow, oh, inv)
        assert np.abs(out - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), (FUNCS[f], inv, (ow, oh))
