"""GPU (-m gpu): zoom's frame at ANY scale, offset and basis by chirp-z transforms (dspfft_zoomczt_*, dspfft_cztrows_*: dct_czt.h) against
the reference's own loop (tests/golden/ref_direct.npz: zoom.c:36-68,361-375 compiled as they lie -- incl. the non-integer scales and the
centered basis the DCT-III grid path refuses), the f64 restatement, the dense MFMA product at BASELINE config 3's size, and the cosine
series itself for the row kernel alone."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dspfun_amd import _lib
    _lib.load()
    return torch


def zoomczt(gpu, coeffs, typ, xn, xd, yn, yd, vx, vy, vw, vh):
    from dspfun_amd import _lib
    L = _lib.load()
    h, w, _ = coeffs.shape
    z = C.c_void_p()
    rc = L.dspfft_zoomczt_create(C.byref(z), w, h, typ, xn, xd, yn, yd, vw, vh)
    if rc:
        return rc, None
    try:
        c = gpu.from_numpy(np.ascontiguousarray(coeffs, dtype=np.float32)).to("cuda:0")
        out = gpu.empty((vh, vw, 3), dtype=gpu.float32, device="cuda:0")
        work = gpu.empty(L.dspfft_zoomczt_work_floats(z), dtype=gpu.float32, device="cuda:0")
        assert L.dspfft_zoomczt_execute(z, c.data_ptr(), vx, vy, out.data_ptr(), work.data_ptr(), None) == 0, L.dspfft_zoomfft_last_error()
        gpu.cuda.synchronize()
        return 0, out.cpu().numpy()
    finally:
        L.dspfft_zoomczt_destroy(z)


def test_against_the_references_compiled_loop_every_case(gpu):
    """all eleven cases of the fixture: three bases, integer / rational / down-scales, offsets -- none refused"""
    fx = np.load(os.path.join(HERE, "golden", "ref_direct.npz"))
    kinds = set()
    for i, case in enumerate(fx["zoom_cases"]):
        h, w, typ = int(case[0]), int(case[1]), int(case[2])
        xn, xd, yn, yd, vx, vy = (float(v) for v in case[3:])
        want = fx[f"zoom{i}_out"]
        vh, vw, _ = want.shape
        rc, got = zoomczt(gpu, fx[f"zoom{i}_coeffs"], typ, xn, xd, yn, yd, vx, vy, vw, vh)
        assert rc == 0, case
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), (i, case)
        integer = abs(w * xn / xd - round(w * xn / xd)) < 1e-9 and abs(h * yn / yd - round(h * yn / yd)) < 1e-9
        kinds.add((typ, integer))
    assert {k[0] for k in kinds} == {0, 1, 2}                  # (non-integer scaled lengths: test_czt_equals_oracle_and_dense_product)


def oracle_frame(x, typ, xs, ys, vx, vy, vw, vh):
    h, w, _ = x.shape
    L = ol.lib()
    cf = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10, impl="port", threads=8))
    cw = L.oracle_zoom_basis_f64(None, typ, xs[0], xs[1], vx, vw, w); ch = L.oracle_zoom_basis_f64(None, typ, ys[0], ys[1], vy, vh, h)
    xb = np.zeros(max(1, vw * (cw - 1))); yb = np.zeros(max(1, vh * (ch - 1)))
    L.oracle_zoom_basis_f64(xb.ctypes.data, typ, xs[0], xs[1], vx, vw, w); L.oracle_zoom_basis_f64(yb.ctypes.data, typ, ys[0], ys[1], vy, vh, h)
    ref = np.zeros((vh, vw, 3))
    L.oracle_zoom_product_f64(cf.ctypes.data, w, h, xb.ctypes.data, cw, yb.ctypes.data, ch, ref.ctypes.data, vw, vh)
    return ref


@pytest.mark.parametrize("w,h,xs,ys,vx,vy,typ,vw,vh", [
    (64, 48, (3.7, 1.0), (3.7, 1.0), 0.0, 0.0, 1, 236, 177),            # centered, non-integer
    (64, 48, (2.83, 1.0), (1.9, 1.0), 7.25, -3.5, 0, 181, 91),          # interpolated off the grid, panned
    (60, 36, (37.0, 10.0), (4.0, 3.0), 0.0, 0.0, 2, 222, 48),           # native, rational
    (96, 64, (0.7, 1.0), (0.5, 1.0), 1.5, 0.0, 0, 67, 32),              # down-scale: fewer components than coefficients
    (200, 120, (4.0, 1.0), (4.0, 1.0), 10.0, 20.0, 1, 500, 300),        # centered at an integer scale, viewport inside the scaled image
    (33, 17, (2.0, 1.0), (2.0, 1.0), 0.5, 0.25, 0, 66, 34),             # on the grid: the same frame the fast path gives
    (640, 360, (3.3, 1.0), (3.3, 1.0), 100.0, 50.0, 1, 1500, 900),      # P = 2400 / 1200
])
def test_czt_equals_oracle_and_dense_product(gpu, w, h, xs, ys, vx, vy, typ, vw, vh):
    from dspfun_amd.zoom import Zoom
    x = ol.synth_f32(w * h + 5, w * h * 3).reshape(h, w, 3)
    z = Zoom(gpu, gpu.from_numpy(x).to("cuda:0"))
    got = z.frame(vw, vh, xs, ys, vx, vy, typ, method="czt").cpu().numpy()
    dense = z.frame(vw, vh, xs, ys, vx, vy, typ, method="gemm").cpu().numpy()
    assert np.abs(got - dense).max() <= 2e-5 * max(1.0, np.abs(dense).max())
    ref = oracle_frame(x, typ, xs, ys, vx, vy, vw, vh)
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_auto_picks_the_grid_path_then_chirp_z_then_the_dense_product(gpu):
    from dspfun_amd.zoom import Zoom
    x = ol.synth_f32(77, 64 * 48 * 3).reshape(48, 64, 3)
    z = Zoom(gpu, gpu.from_numpy(x).to("cuda:0"))
    z.frame(192, 144, (3.0, 1.0), (3.0, 1.0))
    assert len(getattr(z, "_fft", {})) == 1 and not getattr(z, "_czt", {})           # integer scaled lengths: the DCT-III grid
    z.frame(236, 177, (3.7, 1.0), (3.7, 1.0), basis_type=1)
    assert len(z._czt) == 1 and list(z._czt.values())[0] is not None                   # centered: chirp-z


def test_c3_size_centered_3p7(gpu):
    """1920x1080 -> 3.7x, centered basis: chirp-z against the dense product over the whole frame, sampled rows against
    the f64 restatement; timing line (informative)."""
    from dspfun_amd.zoom import Zoom
    w, h = 1920, 1080
    x = ol.synth_f32(0xD5F0003, w * h * 3).reshape(h, w, 3)
    z = Zoom(gpu, gpu.from_numpy(x).to("cuda:0"))
    vw, vh = int(w * 3.7), int(h * 3.7)
    args = (vw, vh, (3.7, 1.0), (3.7, 1.0), 12.5, -4.25, 1)
    got = z.frame(*args, method="czt")
    dense = z.frame(*args, method="gemm")
    gpu.cuda.synchronize()
    assert float((got - dense).abs().max()) <= 1e-5 * float(dense.abs().max())     # north_star: 1e-5 relative (whole frame: tests/test_zoom_c3_tolerance_gpu.py)
    L = ol.lib()
    cf = ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10, impl="port", threads=8)
    cw = L.oracle_zoom_basis_f64(None, 1, 3.7, 1.0, 12.5, vw, w)
    xb = np.zeros(vw * (cw - 1)); L.oracle_zoom_basis_f64(xb.ctypes.data, 1, 3.7, 1.0, 12.5, vw, w)
    XB = np.concatenate([np.full((vw, 1), 0.5), xb.reshape(vw, cw - 1)], axis=1)
    for j in (0, 1777, vh - 1):
        k = (j - 4.25) * (h - 1) / (h * 3.7 - 1)
        ybj = np.cos(np.pi * (k + 0.5) * np.arange(1, h) / h)
        trow = cf[0] / 2 + np.tensordot(ybj, cf[1:], axes=(0, 0))
        ref_row = (XB @ trow[:cw]) / (w * h)
        assert np.abs(got[j].cpu().numpy() - ref_row).max() <= 1e-5 * np.abs(ref_row).max()
    for name in ("czt", "gemm"):
        for _ in range(30):
            z.frame(*args, method=name)
        a, b = gpu.cuda.Event(enable_timing=True), gpu.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            z.frame(*args, method=name)
        b.record(); gpu.cuda.synchronize()
        print(f"1080p -> 3.7x centered frame, {name}: {a.elapsed_time(b) / 30:.3f} ms")


@pytest.mark.parametrize("nc,nout,lines,group", [(1920, 7680, 48, 3), (1080, 4320, 40, 1), (3000, 9000, 6, 3), (5000, 14000, 3, 3), (10, 100, 7, 1)])
def test_cztrows_against_the_series(gpu, nc, nout, lines, group):
    """the row kernel alone on every stage count and the longest listed convolutions (P = 9600, 5400, 12000, 19200, 1200)"""
    from dspfun_amd import _lib
    L = _lib.load()
    omega, phi = np.pi * 0.27 / nc, np.pi * (0.27 * 3.5 + 0.5) / nc
    rng = np.random.default_rng(nc)
    x = rng.random((lines, nc), dtype=np.float32) - np.float32(0.5)
    src = gpu.from_numpy(x).to("cuda:0")
    dst = gpu.full((lines, nout), -77.0, dtype=gpu.float32, device="cuda:0")
    p = C.c_void_p()
    assert L.dspfft_cztrows_create(C.byref(p), nc, nout, lines, group) == 0, L.dspfft_last_error()
    try:
        if group == 1:
            rc = L.dspfft_cztrows_execute(p, src.data_ptr(), nc, 0, 1, dst.data_ptr(), nout, 0, 1, omega, phi, 0.5, None)
        else:      # the members of a group are consecutive planar lines here
            rc = L.dspfft_cztrows_execute(p, src.data_ptr(), 3 * nc, nc, 1, dst.data_ptr(), 3 * nout, nout, 1, omega, phi, 0.5, None)
        assert rc == 0, L.dspfft_last_error()
        gpu.cuda.synchronize()
    finally:
        L.dspfft_cztrows_destroy(p)
    got = dst.cpu().numpy()
    n = np.arange(nc)[None, :]
    for b in (0, 1, nout // 3, nout - 1):
        basis = np.cos(n * (omega * b + phi)); basis[:, 0] *= 0.5
        ref = 0.5 * (x.astype(np.float64) * basis).sum(axis=1)
        assert np.abs(got[:, b] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max() * 30)
    rows = [0, lines - 1]
    bb = np.arange(nout)[:, None]
    full = np.cos(np.arange(nc)[None, :] * (omega * bb + phi)); full[:, 0] *= 0.5
    ref = 0.5 * (x[rows].astype(np.float64) @ full.T)
    assert np.abs(got[rows] - ref).max() <= 1e-5 * np.abs(ref).max()
