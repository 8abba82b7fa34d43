"""GPU (-m gpu): zoom's frame by fast transforms (dspfft_zoomfft_*: two REDFT01 executions per axis, zoom_fft.hip) against the
reference's own loop (tests/golden/ref_direct.npz: zoom.c:36-68,361-375 compiled as they lie), against the oracle restatement and
against the dense MFMA product, incl. BASELINE config 3 at full size."""
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


def zoomfft(gpu, coeffs, typ, xn, xd, yn, yd, vx, vy, vw, vh):
    """-> (rc of create, output or None)"""
    from dspfun_amd import _lib
    L = _lib.load()
    h, w, _ = coeffs.shape
    z = C.c_void_p()
    rc = L.dspfft_zoomfft_create(C.byref(z), w, h, typ, xn, xd, yn, yd, vw, vh)
    if rc:
        return rc, None
    try:
        c = gpu.from_numpy(np.ascontiguousarray(coeffs, dtype=np.float32)).to("cuda:0")
        out = gpu.empty((vh, vw, 3), dtype=gpu.float32, device="cuda:0")
        work = gpu.empty(L.dspfft_zoomfft_work_floats(z), dtype=gpu.float32, device="cuda:0")
        assert L.dspfft_zoomfft_execute(z, c.data_ptr(), vx, vy, out.data_ptr(), work.data_ptr(), None) == 0, L.dspfft_zoomfft_last_error()
        gpu.cuda.synchronize()
        return 0, out.cpu().numpy()
    finally:
        L.dspfft_zoomfft_destroy(z)


def test_against_the_references_compiled_loop(gpu):
    fx = np.load(os.path.join(HERE, "golden", "ref_direct.npz"))
    ran = 0
    for i, case in enumerate(fx["zoom_cases"]):
        h, w, typ = int(case[0]), int(case[1]), int(case[2])
        xn, xd, yn, yd, vx, vy = (float(v) for v in case[3:])
        want = fx[f"zoom{i}_out"]
        vh, vw, _ = want.shape
        rc, got = zoomfft(gpu, fx[f"zoom{i}_coeffs"], typ, xn, xd, yn, yd, vx, vy, vw, vh)
        integer = abs(w * xn / xd - round(w * xn / xd)) < 1e-9 and abs(h * yn / yd - round(h * yn / yd)) < 1e-9
        if typ == 1 or not integer:
            assert rc == -2, case                       # centered basis / non-integer scaled length: the dense product's job
            continue
        assert rc == 0, case
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), case
        ran += 1
    assert ran >= 7


@pytest.mark.parametrize("w,h,xs,ys,vx,vy,typ,vw,vh", [
    (64, 48, (3, 1), (3, 1), 0.0, 0.0, 0, None, None), (64, 48, (2, 1), (5, 2), 7.25, -3.5, 0, None, None), (60, 36, (3, 2), (4, 3), 0.0, 0.0, 2, None, None),
    (60, 36, (1, 2), (1, 3), 1.5, 0.0, 0, None, None), (96, 64, (4, 1), (4, 1), 10.0, 20.0, 0, 100, 60), (33, 17, (2, 1), (2, 1), 0.5, 0.25, 2, None, None),
    # scaled widths with a listed row kernel (256, 640, 1280 RGB pixels): the x stage runs LAST, as row passes over compact windowed lines
    (128, 40, (2, 1), (3, 1), 0.0, 0.0, 0, None, None), (160, 30, (4, 1), (2, 1), 3.25, -1.5, 0, 500, 50), (320, 24, (4, 1), (1, 1), 0.0, 0.0, 2, None, None),
    (320, 18, (2, 1), (5, 2), -2.0, 4.0, 2, 640, 20), (1, 7, (256, 1), (3, 1), 0.0, 0.0, 0, None, None)])
def test_fft_equals_dense_product_and_oracle(gpu, w, h, xs, ys, vx, vy, typ, vw, vh):
    from dspfun_amd.zoom import Zoom
    x = ol.synth_f32(w * h + 5, w * h * 3).reshape(h, w, 3)
    vw = vw or int(w * xs[0] / xs[1]); vh = vh or int(h * ys[0] / ys[1])
    z = Zoom(gpu, gpu.from_numpy(x).to("cuda:0"))
    fast = z.frame(vw, vh, xs, ys, vx, vy, typ, method="fft").cpu().numpy()
    dense = z.frame(vw, vh, xs, ys, vx, vy, typ, method="gemm").cpu().numpy()
    assert np.abs(fast - dense).max() <= 2e-5 * max(1.0, np.abs(dense).max())
    # f64 restatement of zoom.c:36-68,361-375
    L = ol.lib()
    cf = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10))
    cw = L.oracle_zoom_basis_f64(None, typ, xs[0], xs[1], vx, vw, w); ch = L.oracle_zoom_basis_f64(None, typ, ys[0], ys[1], vy, vh, h)
    xb = np.zeros(max(1, vw * (cw - 1))); yb = np.zeros(max(1, vh * (ch - 1)))
    L.oracle_zoom_basis_f64(xb.ctypes.data, typ, xs[0], xs[1], vx, vw, w); L.oracle_zoom_basis_f64(yb.ctypes.data, typ, ys[0], ys[1], vy, vh, h)
    ref = np.zeros((vh, vw, 3))
    L.oracle_zoom_product_f64(cf.ctypes.data, w, h, xb.ctypes.data, cw, yb.ctypes.data, ch, ref.ctypes.data, vw, vh)
    assert np.abs(fast - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_what_does_not_qualify_is_refused(gpu):
    c = np.zeros((12, 16, 3))
    assert zoomfft(gpu, c, 1, 2, 1, 2, 1, 0, 0, 32, 24)[0] == -2          # centered
    assert zoomfft(gpu, c, 0, 5, 3, 2, 1, 0, 0, 26, 24)[0] == -2          # 16 * 5 / 3 is not an integer
    assert zoomfft(gpu, c, 0, 2, 1, 2, 1, 0, 0, 40, 24)[0] == -2          # viewport wider than the scaled image


def test_c3_zoom_4x_1080p_fft(gpu):
    """BASELINE config 3 on the fast path: 1920x1080 RGB -> 7680x4320, scale 4/1, offset 0, interpolated basis.  out[::4, ::4] == input
    (SURVEY 8d), one full output row against the f64 restatement, and agreement with the dense MFMA product."""
    from dspfun_amd.zoom import Zoom
    w, h = 1920, 1080
    x = ol.synth_f32(0xD5F0003, w * h * 3).reshape(h, w, 3)
    z = Zoom(gpu, gpu.from_numpy(x).to("cuda:0"))
    out = z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), method="fft")
    gpu.cuda.synchronize()
    assert np.abs(out[::4, ::4].cpu().numpy() - x).max() <= 1e-5 * np.abs(x).max()     # north_star: 1e-5 relative (whole frame: tests/test_zoom_c3_tolerance_gpu.py)
    dense = z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), method="gemm")
    assert float((out - dense).abs().max()) <= 1e-5 * float(dense.abs().max())
    L = ol.lib()
    cf = ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10, impl="port", threads=8)
    cw = L.oracle_zoom_basis_f64(None, 0, 4.0, 1.0, 0.0, 4 * w, w)
    xb = np.zeros(4 * w * (cw - 1)); L.oracle_zoom_basis_f64(xb.ctypes.data, 0, 4.0, 1.0, 0.0, 4 * w, w)
    for j in (1234, 4319, 1):
        ybj = np.cos(np.pi * ((j / 4.0) + 0.5) * np.arange(1, h) / h)
        trow = cf[0] / 2 + np.tensordot(ybj, cf[1:], axes=(0, 0))
        XB = np.concatenate([np.full((4 * w, 1), 0.5), xb.reshape(4 * w, cw - 1)], axis=1)
        ref_row = (XB @ trow) / (w * h)
        assert np.abs(out[j].cpu().numpy() - ref_row).max() <= 1e-5 * np.abs(ref_row).max()
    # timing (informative): events around 5 frames of each path
    for name in ("fft", "gemm"):
        a, b = gpu.cuda.Event(enable_timing=True), gpu.cuda.Event(enable_timing=True)
        z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), method=name)
        a.record()
        for _ in range(5):
            z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), method=name)
        b.record(); gpu.cuda.synchronize()
        print(f"C3 zoom frame, {name}: {a.elapsed_time(b) / 5:.3f} ms")


def test_input_window_skips_rows_that_are_zero_by_contract(gpu):
    """dspfft_plan_set_input_window: a listed column REDFT01 pass does not read rows outside the window (they may hold garbage); plans
    that cannot honour it say so (return 0) and read everything."""
    from dspfun_amd import Plan, REDFT10, REDFT01
    N, inner = 1080, 64
    x = ol.synth_f32(77, N * inner).reshape(N, inner) - 0.5
    for (lo, hi) in ((0, 270), (811, 1080), (100, 101)):
        pad = np.zeros_like(x); pad[lo:hi] = x[lo:hi]
        want = ol.r2r_many(pad.astype(np.float64), [N], [ol.REDFT01], howmany=inner, istride=inner, idist=1, ostride=inner, odist=1, impl="port").reshape(N, inner)
        p = Plan.guru([(N, inner, inner)], [(inner, 1, 1)], [REDFT01])
        assert "COL*" in p.describe(), p.describe()
        assert p.set_input_window(0, lo, hi) is True
        dirty = x.copy(); dirty[:lo] = np.nan; dirty[hi:] = np.inf          # garbage where the contract says zero
        d = gpu.from_numpy(np.ascontiguousarray(dirty, dtype=np.float32)).to("cuda:0")
        p.execute(d.data_ptr())
        gpu.cuda.synchronize()
        got = d.cpu().numpy()
        assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), (lo, hi)
        assert p.set_input_window(0, 0, 0) is False                           # off again: everything is read
    # not honoured: forward kind, a row pass first, double precision -- the call says so and changes nothing
    assert Plan.guru([(N, inner, inner)], [(inner, 1, 1)], [REDFT10]).set_input_window(0, 0, 270) is False
    assert Plan.image(1080, 1920, 3, REDFT01).set_input_window(0, 0, 270) is False
    assert Plan.guru([(N, inner, inner)], [(inner, 1, 1)], [REDFT01], dtype="f64").set_input_window(0, 0, 270) is False
    from dspfun_amd.engine import DspfftError
    with pytest.raises(DspfftError):
        Plan.guru([(N, inner, inner)], [(inner, 1, 1)], [REDFT01]).set_input_window(0, 5, 2000)


def test_output_alternate_fused_into_the_column_pass(gpu):
    """dspfft_plan_set_output_alternate: output row j times (-1)^j in the pass's store, alone and together with an accumulating execution
    (acc += plan(in)): what zoom's sine part uses instead of a combine kernel"""
    from dspfun_amd import Plan, REDFT01
    N, inner = 1080, 64
    x = ol.synth_f32(78, N * inner).reshape(N, inner) - 0.5
    want = ol.r2r_many(x.astype(np.float64), [N], [ol.REDFT01], howmany=inner, istride=inner, idist=1, ostride=inner, odist=1, impl="port").reshape(N, inner)
    sign = np.where(np.arange(N) % 2 == 1, -1.0, 1.0)[:, None]
    p = Plan.guru([(N, inner, inner)], [(inner, 1, 1)], [REDFT01]).set_scale(-0.25)
    assert p.set_output_alternate(0) is True
    d = gpu.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to("cuda:0")
    out = gpu.empty_like(d)
    p.execute(d.data_ptr(), out.data_ptr())
    gpu.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - (-0.25) * sign * want).max() <= 1e-5 * np.abs(want).max()
    acc0 = ol.synth_f32(79, N * inner).reshape(N, inner)
    acc = gpu.from_numpy(acc0.copy()).to("cuda:0")
    work = gpu.empty_like(d)
    p.execute_masked_accumulate(d.data_ptr(), work.data_ptr(), acc.data_ptr())
    gpu.cuda.synchronize()
    assert np.abs(acc.cpu().numpy() - (acc0 + (-0.25) * sign * want)).max() <= 1e-5 * np.abs(want).max()
    assert p.set_output_alternate(0, False) is False
    assert Plan.image(1080, 1920, 3, REDFT01).set_output_alternate(1) is False         # the x axis runs as a row pass (and first): not honoured


@pytest.mark.parametrize("N,lines,cw", [(640, 7, 160), (7680, 5, 1920), (256, 3, 255)])
def test_row_window_modulation_and_sum_of_two(gpu, N, lines, cw):
    """the pieces zoom's x stage is made of, on listed ROW REDFT01 kernels: input window with compact lines, the mirrored modulated read
    (dspfft_plan_set_input_modulation), the alternating store, and both parts in one launch (dspfft_execute_sum2) against one after the other"""
    from dspfun_amd import Plan, REDFT01
    c = 3
    lo = N - cw + 1
    T = np.ascontiguousarray(ol.synth_f32(N + cw, lines * cw * c).reshape(lines, cw, c) - 0.5, dtype=np.float32)
    ma, mb = ol.synth_f32(7, cw).astype(np.float32), ol.synth_f32(8, cw).astype(np.float32)
    fa = np.zeros((lines, N, c)); fa[:, :cw] = T * ma[None, :, None]
    fb = np.zeros((lines, N, c)); fb[:, lo:] = (T * mb[None, :, None])[:, N - np.arange(lo, N)]
    tr = lambda f: np.stack([ol.r2r_many(f[j], [N], [ol.REDFT01], howmany=c, istride=c, idist=1, ostride=c, odist=1, impl="port").reshape(N, c) for j in range(lines)])
    wa, wb = tr(fa), tr(fb)
    sign = np.where(np.arange(N) % 2 == 1, -1.0, 1.0)[None, :, None]
    want = 0.5 * wa - 0.25 * sign * wb
    qa = Plan.guru([(N, c, c)], [(c, 1, 1), (lines, cw * c, N * c)], [REDFT01]).set_scale(0.5)
    qb = Plan.guru([(N, c, c)], [(c, 1, 1), (lines, cw * c, N * c)], [REDFT01]).set_scale(-0.25)
    assert "ROW*" in qa.describe(), qa.describe()
    dT, dma, dmb = (gpu.from_numpy(a).to("cuda:0") for a in (T, ma, mb))
    assert qa.set_input_window(0, 0, cw) and qa.set_input_modulation(0, dma.data_ptr())
    assert qb.set_input_window(0, lo, N) and qb.set_output_alternate(0) and qb.set_input_modulation(0, dmb.data_ptr(), N)
    both = gpu.full((lines, N, c), float("nan"), dtype=gpu.float32, device="cuda:0")
    qa.execute_sum2(qb, dT.data_ptr(), dT.data_ptr(), both.data_ptr())
    seq = gpu.full((lines, N, c), float("nan"), dtype=gpu.float32, device="cuda:0")
    qa.execute(dT.data_ptr(), seq.data_ptr())
    qb.execute_masked_accumulate(dT.data_ptr(), seq.data_ptr(), seq.data_ptr())
    gpu.cuda.synchronize()
    tol = 1e-5 * (np.abs(wa).max() + np.abs(wb).max())
    assert np.abs(both.cpu().numpy() - want).max() <= tol and np.abs(seq.cpu().numpy() - want).max() <= tol


@pytest.mark.parametrize("N,inner,ch,pitch", [(1080, 64, 270, 96), (4320, 96, 1079, 5760), (2160, 32, 2, 32)])
def test_column_window_modulation_and_mirror(gpu, N, inner, ch, pitch):
    """a listed COL REDFT01 pass reads the rows of its window from (mirrored) rows of a shorter array with another row pitch, times a
    per-row table (dspfft_plan_set_input_modulation): zoom's y stage on the coefficients themselves"""
    from dspfun_amd import Plan, REDFT01
    C0 = np.ascontiguousarray(ol.synth_f32(N + ch, ch * pitch).reshape(ch, pitch) - 0.5, dtype=np.float32)
    ma, mb = ol.synth_f32(22, ch).astype(np.float32), ol.synth_f32(23, ch).astype(np.float32)
    fa = np.zeros((N, inner)); fa[:ch] = C0[:, :inner] * ma[:, None]
    lo = N - ch + 1
    fb = np.zeros((N, inner)); ys = np.arange(lo, N); fb[ys] = (C0[:, :inner] * mb[:, None])[N - ys]
    tr = lambda f: ol.r2r_many(f, [N], [ol.REDFT01], howmany=inner, istride=inner, idist=1, ostride=inner, odist=1, impl="port").reshape(N, inner)
    wa, wb = tr(fa), tr(fb)
    pa = Plan.guru([(N, pitch, inner)], [(inner, 1, 1)], [REDFT01]).set_scale(0.5)
    pb = Plan.guru([(N, pitch, inner)], [(inner, 1, 1)], [REDFT01]).set_scale(-0.5)
    assert "COL*" in pa.describe(), pa.describe()
    dC, dma, dmb = (gpu.from_numpy(a).to("cuda:0") for a in (C0, ma, mb))
    assert pa.set_input_window(0, 0, ch) and pa.set_input_modulation(0, dma.data_ptr())
    assert pb.set_input_window(0, lo, N) and pb.set_input_modulation(0, dmb.data_ptr(), N) and pb.set_output_alternate(0)
    out = gpu.full((N, inner), float("nan"), dtype=gpu.float32, device="cuda:0")
    work = gpu.zeros((N, inner), dtype=gpu.float32, device="cuda:0")
    pa.execute(dC.data_ptr(), out.data_ptr())
    pb.execute_masked_accumulate(dC.data_ptr(), work.data_ptr(), out.data_ptr())
    gpu.cuda.synchronize()
    sign = np.where(np.arange(N) % 2 == 1, -1.0, 1.0)[:, None]
    want = 0.5 * wa - 0.5 * sign * wb
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-5 * (np.abs(wa).max() + np.abs(wb).max())
