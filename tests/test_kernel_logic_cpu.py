"""CPU: the planner (engine.cpp) and the kernel phase functions (dct_core.h) run through the
test-only emulation backend and are compared with the oracle.  This is NOT the parity test of the
product (that is tests/test_gpu_parity.py, -m gpu, through the HIP library); it checks the logic
that both builds share, in a container without a GPU."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from dspfun_amd.engine import Plan, DspfftError, REDFT10, REDFT01
from emul_lib import emul

TOL = 2e-6   # f32 FFT-based vs f64 definition, relative to max|ref|


def run(plan, x, out=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    if out is None:
        plan.execute(x.ctypes.data)
        return x
    plan.execute(x.ctypes.data, out.ctypes.data)
    return out


def relerr(got, ref):
    return np.abs(got.astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30)


@pytest.mark.parametrize("h,w,c", [(48, 64, 3), (16, 16, 3), (30, 60, 3), (54, 120, 3), (8, 2, 3), (2, 8, 1), (60, 90, 1),
                                    (15, 27, 3), (7, 13, 2), (9, 10, 4), (1, 8, 3), (8, 1, 3), (45, 50, 2)])
@pytest.mark.parametrize("kind", [REDFT10, REDFT01])
def test_image_plan_inplace(h, w, c, kind):
    x = ol.synth_f32(h * 1000 + w, h * w * c).reshape(h, w, c)
    ref = ol.dct2d_interleaved(x.astype(np.float64), kind, impl="port")
    p = Plan.image(h, w, c, kind, lib=emul())
    got = run(p, x.copy())
    assert relerr(got, ref) < TOL, p.describe()


def test_describe_picks_row_and_col():
    p = Plan.image(48, 64, 3, REDFT10, lib=emul())
    d = p.describe()
    assert "ROW" in d and "COL" in d and "DENSE" not in d
    p = Plan.image(37, 40, 3, REDFT10, lib=emul())   # 37 is a prime > 13 (and too long for the in-register pass) -> Bluestein along y, x still direct
    d = p.describe()
    assert "BLUE" in d and "conv=75" in d and "ROW" in d and "DENSE" not in d
    assert "TINY" in Plan.image(17, 40, 3, REDFT10, lib=emul()).describe()      # up to 32: one line per thread, any factorisation
    p = Plan.image(8, 1366, 3, REDFT10, lib=emul())   # 1366 = 2 x 683: 2880 = 12x15x16 beats the smaller 2744 = 7x7x7x8
    assert "BLUE N=1366" in p.describe() and "conv=2880" in p.describe()
    os.environ["DSPFFT_NO_BLUESTEIN"] = "1"
    try:
        d = Plan.image(37, 40, 3, REDFT10, lib=emul()).describe()
        assert "DENSE" in d and "ROW" in d
    finally:
        del os.environ["DSPFFT_NO_BLUESTEIN"]


@pytest.mark.parametrize("kind", [REDFT10, REDFT01])
def test_out_of_place_scan_plan(kind):
    # scan/scan.c:359: reconstruction -> image, input must stay untouched
    h, w, c = 24, 40, 3
    x = ol.synth_f32(77, h * w * c).reshape(h, w, c)
    keep = x.copy()
    out = np.full_like(x, np.nan)
    p = Plan.image(h, w, c, kind, lib=emul())
    run(p, x, out)
    assert np.array_equal(x, keep)
    assert relerr(out, ol.dct2d_interleaved(x.astype(np.float64), kind, impl="port")) < TOL


@pytest.mark.parametrize("N", [2, 4, 6, 8, 10, 12, 16, 18, 20, 24, 30, 32, 36, 48, 60, 64, 90, 120, 128, 240, 256, 270, 540, 1080, 1920, 2160, 3840,
                               1, 3, 5, 7, 9, 15, 17, 19, 27, 31, 45, 75, 77, 97, 135])
def test_1d_lengths_row_and_strided(N):
    x = ol.synth_f32(N, N)
    for kind in (REDFT10, REDFT01):
        ref = ol.r2r_many(x.astype(np.float64), [N], [kind], impl="port")
        p = Plan.many_r2r([N], [kind], lib=emul())
        assert relerr(run(p, x.copy()), ref) < TOL, (N, p.describe())
    # batch of 6 strided signals: element j of signal t at j*6 + t  (COL shape)
    xb = ol.synth_f32(N + 5, N * 6).reshape(N, 6)
    for kind in (REDFT10, REDFT01):
        ref = np.stack([ol.r2r_many(xb[:, t].astype(np.float64), [N], [kind], impl="port") for t in range(6)], axis=1)
        p = Plan.many_r2r([N], [kind], howmany=6, istride=6, idist=1, ostride=6, odist=1, lib=emul())
        assert relerr(run(p, xb.copy()), ref) < TOL, (N, p.describe())


def test_volume_embedded_motion_plan(golden):
    # motion/motion.c:535-552: {d,h,w} inside {md,mh,mw}, in place; gaps must be preserved
    d, h, w, md, mh, mw = [int(v) for v in golden["vol_dims"]]
    buf = golden["vol_in"].astype(np.float32)
    for kind, name in ((REDFT10, "redft10"), (REDFT01, "redft01")):
        p = Plan.many_r2r([d, h, w], [kind] * 3, inembed=[md, mh, mw], onembed=[md, mh, mw], lib=emul())
        got = run(p, buf.copy())
        ref = golden[f"vol_{name}"]
        assert relerr(got, ref) < TOL, p.describe()
        mask = np.ones((md, mh, mw), bool); mask[:d, :h, :w] = False
        assert np.array_equal(got[mask], buf[mask])


def test_volume_dense_3d():
    d, h, w = 8, 12, 20
    x = ol.synth_f32(9, d * h * w)
    for kind in (REDFT10, REDFT01):
        ref = ol.r2r_many(x.astype(np.float64), [d, h, w], [kind] * 3, impl="port")
        p = Plan.many_r2r([d, h, w], [kind] * 3, lib=emul())
        assert relerr(run(p, x.copy()), ref) < TOL, p.describe()


def test_planar_batch():
    # howmany planes at dist = h*w (planar), rank 2
    h, w, b = 12, 20, 3
    x = ol.synth_f32(11, b * h * w)
    ref = ol.r2r_many(x.astype(np.float64), [h, w], [REDFT10] * 2, howmany=b, idist=h * w, odist=h * w, impl="port")
    p = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=b, idist=h * w, odist=h * w, lib=emul())
    assert relerr(run(p, x.copy()), ref) < TOL, p.describe()


def test_r2r_2d_draw_plan():
    x = ol.synth_f32(13, 10 * 14)
    p = Plan.r2r_2d(10, 14, REDFT01, REDFT01, lib=emul())
    ref = ol.r2r_many(x.astype(np.float64), [10, 14], [REDFT01] * 2, impl="port")
    assert relerr(run(p, x.copy()), ref) < TOL


def test_roundtrip_and_fused_spec_normalisation():
    # spec.c:63-78 as one fused plan, then ispec.c:153-167 as one fused plan == identity
    h, w, c = 18, 30, 3
    x = ol.synth_f32(21, h * w * c).reshape(h, w, c)
    r2 = np.float32(np.sqrt(2.0))
    fwd = Plan.image(h, w, c, REDFT10, lib=emul()).set_scale(1.0 / (2 * w * h)).set_axis_scale0(0, 1, 1 / r2).set_axis_scale0(1, 1, 1 / r2)
    inv = Plan.image(h, w, c, REDFT01, lib=emul()).set_scale(0.5).set_axis_scale0(0, r2, 1).set_axis_scale0(1, r2, 1)
    f = run(fwd, x.copy())
    ref = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), REDFT10))
    ol.lib().oracle_spec_normalise_f64(ref.ctypes.data, w, h, c)
    assert np.abs(f - ref).max() < 1e-6          # values are in [-1, 1]
    y = run(inv, f.copy())
    assert np.abs(y - x).max() < 5e-6


def test_lds_pressure_paths(monkeypatch):
    # with a small LDS budget COL narrows its tile; a line that no longer fits the ROW pass goes to the column pass
    import ctypes as C, os
    os.environ["DSPFFT_EMUL_LDS"] = "1100"   # the 64-pixel RGB line needs 32*3*8 = 768 B: still ROW; the 48-row tile shrinks to K=4
    try:
        h, w, c = 48, 64, 3
        x = ol.synth_f32(31, h * w * c).reshape(h, w, c)
        p = Plan.image(h, w, c, REDFT10, lib=emul())
        assert "ROW" in p.describe() and "K=4" in p.describe()
        assert relerr(run(p, x.copy()), ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port")) < TOL
        os.environ["DSPFFT_EMUL_LDS"] = "700"
        p = Plan.image(h, w, c, REDFT10, lib=emul())
        assert "ROW" not in p.describe() and p.describe().count("COL") == 2, p.describe()
        assert relerr(run(p, x.copy()), ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port")) < TOL
    finally:
        del os.environ["DSPFFT_EMUL_LDS"]


def test_errors():
    with pytest.raises(DspfftError):
        Plan.many_r2r([8, 8, 8, 8], [REDFT10] * 4, lib=emul())
    with pytest.raises(DspfftError):
        Plan.many_r2r([8], [3], lib=emul())


def test_helpers_zigzag_and_scan_step():
    import ctypes as C
    L = emul()
    for (w, h) in [(8, 8), (6, 4), (4, 6), (16, 9), (9, 16), (1, 7), (7, 1)]:
        lin = np.zeros(w * h, dtype=np.uint32)
        assert L.dspfft_scan_zigzag(lin.ctypes.data, w, h, 0, w * h, None) == 0
        assert np.array_equal(lin.astype(np.uint64), ol.zigzag_order(w, h))
    # partial range
    lin = np.zeros(10, dtype=np.uint32)
    assert L.dspfft_scan_zigzag(lin.ctypes.data, 16, 9, 50, 10, None) == 0
    assert np.array_equal(lin.astype(np.uint64), ol.zigzag_order(16, 9)[50:60])


# ---- compile-time-specialised kernels (dct_spec.h / spec_list.h), through the emulation backend ----
@pytest.mark.parametrize("h,w,c", [(4, 3840, 3), (3, 1920, 3), (2, 7680, 3), (5, 960, 3), (6, 256, 3), (4, 1920, 1), (4, 960, 1),
                                    (2160, 8, 3), (1080, 16, 3), (1080, 32, 1), (4320, 8, 1), (4320, 4, 3), (540, 16, 1), (256, 16, 3), (256, 256, 3),
                                    (2, 4096, 3), (3, 2560, 3), (2, 2048, 3), (3, 1280, 3), (3, 1024, 3), (5, 720, 3), (4, 640, 3), (4, 512, 3), (4, 1280, 1),
                                    (4096, 8, 1), (4096, 4, 3), (2048, 8, 1), (1440, 8, 3), (1024, 16, 1), (720, 16, 3), (512, 16, 1), (480, 16, 3),
                                    (2, 5120, 3), (2, 3200, 3), (3, 2880, 3), (3, 1600, 3), (3, 1440, 3), (4, 800, 3), (3, 3840, 1), (3, 2560, 1), (3, 4096, 1),
                                    (4, 2048, 1), (5, 1024, 1), (2880, 8, 1), (1800, 8, 1), (1600, 8, 3), (1200, 16, 1), (1152, 16, 3), (960, 16, 1), (900, 16, 3),
                                    (768, 16, 1), (600, 16, 3)])
@pytest.mark.parametrize("kind", [REDFT10, REDFT01])
def test_specialised_kernels(h, w, c, kind):
    x = ol.synth_f32(h * 7 + w, h * w * c).reshape(h, w, c)
    ref = ol.dct2d_interleaved(x.astype(np.float64), kind, impl="port", threads=4)
    p = Plan.image(h, w, c, kind, lib=emul())
    assert "*" in p.describe(), p.describe()
    got = run(p, x.copy())
    assert relerr(got, ref) < TOL, p.describe()
    # fused scaling on the specialised path
    p.set_scale(0.25).set_axis_scale0(0, 1.5, 0.5).set_axis_scale0(1, 2.0, 0.75)
    xs = x.astype(np.float64).copy()
    xs[0, :, :] *= 1.5
    xs[:, 0, :] *= 2.0
    r2 = ol.dct2d_interleaved(xs, kind, impl="port", threads=4) * 0.25
    r2[0, :, :] *= 0.5
    r2[:, 0, :] *= 0.75
    assert relerr(run(p, x.copy()), r2) < TOL


def test_specialised_out_of_place_and_misaligned_fallback():
    h, w, c = 2160, 8, 3
    x = ol.synth_f32(3, h * w * c + 1)
    xin = x[:-1].reshape(h, w, c)
    out = np.zeros_like(xin)
    p = Plan.image(h, w, c, REDFT01, lib=emul())
    run(p, xin, out)
    ref = ol.dct2d_interleaved(xin.astype(np.float64), REDFT01, impl="port", threads=4)
    assert relerr(out, ref) < TOL
    # a buffer that is only 4-byte aligned must still work (generic kernels take over)
    xm = x[1:].reshape(h, w, c).copy()
    buf = np.zeros(h * w * c + 4, dtype=np.float32)
    view = buf[1:1 + h * w * c]
    view[:] = xm.ravel()
    p.execute(view.ctypes.data)
    assert relerr(view.reshape(h, w, c), ol.dct2d_interleaved(xm.astype(np.float64), REDFT01, impl="port", threads=4)) < TOL


@pytest.mark.parametrize("h,w,c", [(24, 40, 3), (4, 3840, 3), (2160, 8, 3), (2160, 16, 3), (17, 40, 3), (30, 45, 1)])
def test_fused_scan_step(h, w, c):
    """scan/scan.c:429-459 in one fused execution == scatter + REDFT01^2 + accumulate, frame by frame"""
    import ctypes as C
    L = emul()
    x = ol.synth_f32(h * 3 + w, h * w * c).reshape(h, w, c)
    coeffs = x.copy()
    Plan.image(h, w, c, REDFT10, lib=L).set_scale(1.0 / (4 * w * h)).execute(coeffs.ctypes.data)
    ids = np.zeros(h * w, dtype=np.uint32)
    nframes = 5
    step = (h * w + nframes - 1) // nframes
    assert L.dspfft_scan_zigzag_frame_ids(ids.ctypes.data, w, h, step, None) == 0
    zz = ol.zigzag_order(w, h)
    assert ids[0] == 0xFFFFFFFF and np.array_equal(ids[zz[1:].astype(np.int64)], (np.arange(1, h * w) // step).astype(np.uint32))
    inv = Plan.image(h, w, c, REDFT01, lib=L)
    acc = np.ascontiguousarray(np.broadcast_to(coeffs[0, 0], (h, w, c)).copy())
    work = np.zeros_like(acc)
    ref = acc.astype(np.float64).copy()
    c64 = coeffs.astype(np.float64)
    for f in range(nframes):
        inv.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc.ctypes.data, ids.ctypes.data, f, c)
        rec = np.where((ids.reshape(h, w) == f)[:, :, None], c64, 0.0)
        ref += ol.dct2d_interleaved(rec, REDFT01, impl="port")
        assert np.abs(acc - ref).max() < 1e-5, f
    assert np.abs(acc - x).max() < 1e-5


# ---- double-precision plans (the fftw_ API of spec's default build, include/precision.h:50-53) ----
TOL64 = 5e-14   # f64 FFT-based kernels vs the f64 direct-definition restatement, relative to max|ref|


def run64(plan, x, out=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    if out is None:
        plan.execute(x.ctypes.data)
        return x
    plan.execute(x.ctypes.data, out.ctypes.data)
    return out


@pytest.mark.parametrize("h,w,c", [(48, 64, 3), (30, 60, 3), (8, 2, 3), (60, 90, 1), (15, 27, 3), (7, 13, 2), (9, 10, 4), (17, 40, 3), (1, 8, 3)])
@pytest.mark.parametrize("kind", [REDFT10, REDFT01])
def test_f64_image_plan(h, w, c, kind):
    x = ol.synth_f32(h * 1000 + w, h * w * c).astype(np.float64).reshape(h, w, c) + 1e-9 * np.arange(h * w * c).reshape(h, w, c)
    ref = ol.dct2d_interleaved(x, kind)                       # direct definition, f64
    p = Plan.image(h, w, c, kind, lib=emul(), dtype="f64")
    assert "f64" in p.describe()
    got = run64(p, x.copy())
    assert relerr(got, ref) < TOL64, p.describe()
    # the f32 plan of the same geometry cannot get there: the f64 path is not a converted f32 path
    p32 = Plan.image(h, w, c, kind, lib=emul())
    got32 = run(p32, x.astype(np.float32))
    if h * w > 16:
        assert relerr(got32, ref) > 100 * TOL64


@pytest.mark.parametrize("kind", [REDFT10, REDFT01])
def test_f64_8k_lines_run_as_channel_lines_only(kind, monkeypatch):
    """a 7680 x 3 double line is 184 KB -- more than a CU's LDS: it has no interleaved kernel, only channel lines (round 4), and those
    run even where DSPFFT_ROW_CHAN=0 would switch channel lines off"""
    monkeypatch.setenv("DSPFFT_ROW_CHAN", "0")
    h, w, c = 5, 7680, 3
    x = ol.synth_f32(77, h * w * c).astype(np.float64).reshape(h, w, c) * (1 + 2.0 ** -31)
    p = Plan.image(h, w, c, kind, lib=emul(), dtype="f64")
    assert "ROW* f64 N=7680 C=3" in p.describe() and "3 channel lines" in p.describe(), p.describe()
    got = run64(p, x.copy())
    ref = ol.dct2d_interleaved(x, kind, impl="port")
    assert relerr(got, ref) < TOL64


@pytest.mark.parametrize("N", [2, 6, 16, 30, 64, 270, 1080, 3840, 1, 3, 17, 45, 97, 135])
def test_f64_1d_lengths(N):
    x = ol.synth_f32(N, N).astype(np.float64) * (1 + 2.0 ** -30)
    for kind in (REDFT10, REDFT01):
        ref = ol.r2r_many(x, [N], [kind])
        p = Plan.many_r2r([N], [kind], lib=emul(), dtype="f64")
        assert relerr(run64(p, x.copy()), ref) < TOL64, (N, p.describe())
    xb = ol.synth_f32(N + 5, N * 6).astype(np.float64).reshape(N, 6)
    for kind in (REDFT10, REDFT01):
        ref = np.stack([ol.r2r_many(xb[:, t].copy(), [N], [kind]) for t in range(6)], axis=1)
        p = Plan.many_r2r([N], [kind], howmany=6, istride=6, idist=1, ostride=6, odist=1, lib=emul(), dtype="f64")
        assert relerr(run64(p, xb.copy()), ref) < TOL64, (N, p.describe())


def test_f64_spec_normalisation_roundtrip_and_out_of_place():
    """spec/spec.c:63-78 then spec/ispec.c:153-167 in double, normalisation fused; out-of-place leaves the input alone"""
    h, w, c = 36, 60, 3
    x = ol.synth_f32(5, h * w * c).astype(np.float64).reshape(h, w, c)
    r2 = np.sqrt(2.0)
    fwd = Plan.image(h, w, c, REDFT10, lib=emul(), dtype="f64").set_scale(1.0 / (2.0 * w * h))
    inv = Plan.image(h, w, c, REDFT01, lib=emul(), dtype="f64").set_scale(0.5)
    for a in range(2):
        fwd.set_axis_scale0(a, 1.0, 1.0 / r2)
        inv.set_axis_scale0(a, r2, 1.0)
    f = np.full_like(x, np.nan)
    keep = x.copy()
    run64(fwd, x, f)
    assert np.array_equal(x, keep)
    ref = np.ascontiguousarray(ol.dct2d_interleaved(x, REDFT10))
    ol.lib().oracle_spec_normalise_f64(ref.ctypes.data, w, h, c)
    assert np.abs(f - ref).max() < 1e-15 * 50
    y = run64(inv, f.copy())
    assert np.abs(y - x).max() < 1e-14


def test_f64_fused_scan_step_and_type_mismatch():
    h, w, c = 24, 40, 3
    L = emul()
    x = ol.synth_f32(9, h * w * c).astype(np.float64).reshape(h, w, c)
    coeffs = x.copy()
    Plan.image(h, w, c, REDFT10, lib=L, dtype="f64").set_scale(1.0 / (4 * w * h)).execute(coeffs.ctypes.data)
    ids = np.zeros(h * w, dtype=np.uint32)
    nframes = 4
    step = (h * w + nframes - 1) // nframes
    assert L.dspfft_scan_zigzag_frame_ids(ids.ctypes.data, w, h, step, None) == 0
    inv = Plan.image(h, w, c, REDFT01, lib=L, dtype="f64")
    acc = np.ascontiguousarray(np.broadcast_to(coeffs[0, 0], (h, w, c)).copy())
    work = np.zeros_like(acc)
    for f in range(nframes):
        inv.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc.ctypes.data, ids.ctypes.data, f, c)
    assert np.abs(acc - x).max() < 1e-14
    # an f64 plan refuses the f32 entry point and vice versa
    import ctypes as C
    assert L.dspfft_execute(inv._h, C.c_void_p(acc.ctypes.data), C.c_void_p(acc.ctypes.data), None) != 0
    p32 = Plan.image(h, w, c, REDFT01, lib=L)
    assert L.dspfft_execute_f64(p32._h, C.c_void_p(acc.ctypes.data), C.c_void_p(acc.ctypes.data), None) != 0


# ---- lengths with prime factors > 13: Bluestein's convolution inside the column pass (and the O(N^2) fallback) ----
@pytest.mark.parametrize("h,w,c", [(37, 40, 3), (6, 34, 3), (8, 1366, 3), (97, 6, 1), (5, 683, 1), (41, 41, 1), (43, 47, 2), (1, 37, 3), (53, 1, 1), (3, 2731, 1)])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_bluestein_lengths(h, w, c, dtype):
    for kind in (REDFT10, REDFT01):
        if dtype == "f32":
            x = ol.synth_f32(h * 31 + w, h * w * c).reshape(h, w, c)
            ref = ol.dct2d_interleaved(x.astype(np.float64), kind, impl="port")
            p = Plan.image(h, w, c, kind, lib=emul())
            assert "BLUE" in p.describe() and "DENSE" not in p.describe(), p.describe()
            assert relerr(run(p, x.copy()), ref) < TOL, p.describe()
        else:
            x = ol.synth_f32(h * 31 + w, h * w * c).astype(np.float64).reshape(h, w, c) * (1 + 2.0 ** -31)
            ref = ol.dct2d_interleaved(x, kind, impl="port")
            p = Plan.image(h, w, c, kind, lib=emul(), dtype="f64")
            assert "BLUE f64" in p.describe(), p.describe()
            assert relerr(run64(p, x.copy()), ref) < 2e-13, p.describe()


def test_bluestein_fused_scan_step_and_dense_fallback():
    """the fused masked/accumulating execution runs through the Bluestein pass unchanged; with the path disabled the
    O(N^2) kernel gives the same answer"""
    h, w, c = 37, 74, 3
    L = emul()
    x = ol.synth_f32(4, h * w * c).reshape(h, w, c)
    coeffs = x.copy()
    Plan.image(h, w, c, REDFT10, lib=L).set_scale(1.0 / (4 * w * h)).execute(coeffs.ctypes.data)
    ids = np.zeros(h * w, dtype=np.uint32)
    step = (h * w + 2) // 3
    assert L.dspfft_scan_zigzag_frame_ids(ids.ctypes.data, w, h, step, None) == 0
    inv = Plan.image(h, w, c, REDFT01, lib=L)
    assert inv.describe().count("BLUE") == 2
    acc = np.ascontiguousarray(np.broadcast_to(coeffs[0, 0], (h, w, c)).copy())
    work = np.zeros_like(acc)
    for f in range(3):
        inv.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc.ctypes.data, ids.ctypes.data, f, c)
    assert np.abs(acc - x).max() < 1e-5
    os.environ["DSPFFT_NO_BLUESTEIN"] = "1"
    try:
        pd = Plan.image(h, w, c, REDFT10, lib=L)
    finally:
        del os.environ["DSPFFT_NO_BLUESTEIN"]
    assert pd.describe().count("DENSE") == 2
    a, b = x.copy(), x.copy()
    pd.execute(a.ctypes.data)
    Plan.image(h, w, c, REDFT10, lib=L).execute(b.ctypes.data)
    assert relerr(a, b.astype(np.float64)) < TOL


# ---- pass order and the fused forward -> filter -> inverse column pass (motion/motion.c:641-753) ----
def _oracle_filter(c, active, minbuf_hw, flt):
    import ctypes as C
    O = ol.lib()
    I3, I2 = C.c_int * 3, C.c_int * 2
    O.oracle_motion_filter_f32.restype = C.c_ulonglong
    O.oracle_motion_filter_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_double, C.c_float]
    return O.oracle_motion_filter_f32(c.ctypes.data, I3(*active), I2(*minbuf_hw), I3(*flt["band_begin"]), I3(*flt["band_end"]), flt.get("damp", 1.0), flt.get("boost", 1.0),
                                      flt.get("threshold_lo", 0.0), flt.get("threshold_hi", 0.0), flt.get("preserve_dc", 0), flt.get("grey_add", 0.0), flt.get("quantizer", 0.0))


def test_first_axis_first_order_gives_the_same_transform():
    d, h, w = 6, 10, 16
    x = ol.synth_f32(3, d * h * w)
    for kind in (REDFT10, REDFT01):
        a, b = x.copy(), x.copy()
        pa = Plan.many_r2r([d, h, w], [kind] * 3, lib=emul())
        pb = Plan.many_r2r([d, h, w], [kind] * 3, lib=emul(), first_axis_first=True)
        assert pa.describe().splitlines()[1].startswith("axis 2") and pb.describe().splitlines()[1].startswith("axis 0")
        pa.execute(a.ctypes.data); pb.execute(b.ctypes.data)
        assert relerr(b, a.astype(np.float64)) < 1e-6


@pytest.mark.parametrize("case", ["volume", "frames", "frames_nofilter", "volume_quant"])
def test_fused_roundtrip_matches_unfused_and_oracle(case):
    import ctypes as C
    L = emul()
    r2 = float(np.sqrt(2.0))
    if case.startswith("volume"):
        d, h, w = 256, 4, 8                       # one 3-D block: z is the fused axis (COL* N=256 K=32, inner = h*w = 32: one tile)
        n, howmany, dist, bd = [d, h, w], 1, 0, d
        active, frames = (d, h, w), 1
    else:
        d, h, w = 3, 480, 16                      # per-frame 2-D blocks (motion's default -b 0x0x1): y is the fused axis
        n, howmany, dist, bd = [h, w], d, h * w, 1
        active, frames = (1, h, w), d
    x = (ol.synth_u8(7, d * h * w).astype(np.float32)).reshape(d, h, w)
    rank = len(n)
    nrm = 1.0 / np.prod([2.0 * v for v in n])     # REDFT01(REDFT10(x)) = prod(2 n) x
    fwd = Plan.many_r2r(n, [REDFT10] * rank, howmany=howmany, idist=dist, odist=dist, lib=L).set_scale(2 * r2)
    inv = Plan.many_r2r(n, [REDFT01] * rank, howmany=howmany, idist=dist, odist=dist, lib=L, first_axis_first=True).set_scale(nrm / (2 * r2))
    ref_inv = Plan.many_r2r(n, [REDFT01] * rank, howmany=howmany, idist=dist, odist=dist, lib=L).set_scale(nrm / (2 * r2))
    for a in range(rank):
        fwd.set_axis_scale0(a, 1.0, 1.0 / r2); inv.set_axis_scale0(a, r2, 1.0); ref_inv.set_axis_scale0(a, r2, 1.0)
    assert "COL*" in fwd.describe().splitlines()[-1] and "COL*" in inv.describe().splitlines()[1]
    if case.startswith("volume"):
        assert "N=256 K=32" in fwd.describe().splitlines()[-1]       # round 6: the 128-byte z tile where the inner extent is a multiple of 32
    flt = None
    if case == "volume":
        flt = dict(active=active, minbuf_hw=(h, w), block_depth=bd, band_begin=(0, 1, 2), band_end=(100, 4, 7), damp=0.25, boost=1.5, preserve_dc=1)
    elif case == "volume_quant":
        flt = dict(active=active, minbuf_hw=(h, w), block_depth=bd, band_begin=(0, 0, 0), band_end=active, threshold_lo=2.0, threshold_hi=1e9, preserve_dc=2, grey_add=3.5, quantizer=4.0)
    elif case == "frames":
        flt = dict(active=active, minbuf_hw=(h, w), block_depth=bd, band_begin=(0, 3, 1), band_end=(1, 300, 12), damp=0.5, boost=1.0, quantizer=2.5)
    # reference: forward, the oracle's filter block by block, inverse in the default order
    ref = x.copy()
    fwd.execute(ref.ctypes.data)
    ncoded = 0
    if flt:
        for f in range(frames):
            blk = ref.reshape(frames, -1)[f]
            ncoded += _oracle_filter(blk, active, (h, w), flt)
    ref_inv.execute(ref.ctypes.data)
    # fused
    got = x.copy()
    coded = np.zeros(1, dtype=np.uint64)
    fwd.roundtrip(inv, got.ctypes.data, filter=flt, d_coded=coded.ctypes.data)
    assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()     # u8-range values, two pass orders
    if flt and flt.get("quantizer"):
        assert int(coded[0]) == ncoded
    if not flt:
        assert np.abs(got - x).max() < 2e-3
    # unfused execution of the same plan pair: bit-identical
    os.environ["DSPFFT_NO_FUSED_ROUNDTRIP"] = "1"
    try:
        got2 = x.copy()
        coded2 = np.zeros(1, dtype=np.uint64)
        fwd.roundtrip(inv, got2.ctypes.data, filter=flt, d_coded=coded2.ctypes.data)
    finally:
        del os.environ["DSPFFT_NO_FUSED_ROUNDTRIP"]
    assert np.array_equal(got, got2) and coded[0] == coded2[0]


@pytest.mark.parametrize("case", ["volume", "frames", "fallback"])
def test_roundtrip_u8_matches_float_path(case):
    """motion's 8-bit ends (motion.c:617-640, :760-776) fused into the planar row passes: identical bytes to
    u8 -> float, float roundtrip, dspfft_f32_to_u8"""
    L = emul()
    if case == "volume":
        d, h, w = 256, 2, 960
        n, howmany, dist, bd, active = [d, h, w], 1, 0, d, (d, h, w)
    elif case == "frames":
        d, h, w = 2, 480, 960
        n, howmany, dist, bd, active = [h, w], d, h * w, 1, (1, h, w)
    else:
        d, h, w = 2, 480, 48                      # no specialised row kernel for 48-sample rows: conversions run as separate sweeps
        n, howmany, dist, bd, active = [h, w], d, h * w, 1, (1, h, w)
    rank = len(n)
    u8 = ol.synth_u8(21, d * h * w)
    nrm = 1.0 / np.prod([2.0 * v for v in n])
    fwd = Plan.many_r2r(n, [REDFT10] * rank, howmany=howmany, idist=dist, odist=dist, lib=L)
    inv = Plan.many_r2r(n, [REDFT01] * rank, howmany=howmany, idist=dist, odist=dist, lib=L, first_axis_first=True).set_scale(nrm)
    assert ("ROW*" in fwd.describe().splitlines()[1]) == (case != "fallback")
    flt = dict(active=active, minbuf_hw=(h, w), block_depth=bd, band_begin=(0, 0, 0), band_end=active, quantizer=6.0)
    mul = 0.97
    # float path
    f = u8.astype(np.float32)
    fwd.roundtrip(inv, f.ctypes.data, filter=flt)
    ref = np.zeros(d * h * w, dtype=np.uint8)
    assert L.dspfft_f32_to_u8(ref.ctypes.data, f.ctypes.data, mul, d * h * w, None) == 0
    # 8-bit path
    out = np.zeros(d * h * w, dtype=np.uint8)
    work = np.full(d * h * w, np.nan, dtype=np.float32)
    fwd.roundtrip_u8(inv, u8.ctypes.data, out.ctypes.data, work.ctypes.data, mul, filter=flt)
    assert np.array_equal(out, ref)
    # and it is the right answer: the quantiser is the only loss
    assert np.abs(out.astype(np.float64) - np.clip(np.floor(u8 * mul + 0.5), 0, 255)).max() <= 6


@pytest.mark.parametrize("shape", ["rank1", "image_fused", "image_generic", "out_of_place"])
def test_roundtrip_plan_shapes(shape):
    """dspfft_execute_roundtrip on plan pairs other than motion's: rank 1 (nothing to fuse), interleaved images with and
    without a specialised column kernel, and out of place (the forward plan reads d_in, everything else works on d_out)"""
    L = emul()
    if shape == "rank1":
        n, kw = [96], {}
        x = ol.synth_f32(1, 96)
    elif shape == "image_fused":
        h, w, c = 256, 16, 3
        n, kw = [h, w], dict(howmany=c, istride=c, idist=1, ostride=c, odist=1)
        x = ol.synth_f32(2, h * w * c)
    else:
        h, w, c = 30, 20, 3
        n, kw = [h, w], dict(howmany=c, istride=c, idist=1, ostride=c, odist=1)
        x = ol.synth_f32(3, h * w * c)
    nrm = 1.0 / np.prod([2.0 * v for v in n])
    fwd = Plan.many_r2r(n, [REDFT10] * len(n), lib=L, **kw)
    inv = Plan.many_r2r(n, [REDFT01] * len(n), lib=L, first_axis_first=True, **kw).set_scale(nrm)
    if shape == "image_fused":
        assert "COL*" in fwd.describe().splitlines()[-1]
    if shape == "out_of_place":
        src, dst = x.copy(), np.full_like(x, np.nan)
        fwd.roundtrip(inv, src.ctypes.data, dst.ctypes.data)
        assert np.array_equal(src, x)
        got = dst
    else:
        got = x.copy()
        fwd.roundtrip(inv, got.ctypes.data)
    assert np.abs(got - x).max() < 5e-6
    # a plan pair whose middle axes differ is refused
    if len(n) == 2:
        bad = Plan.many_r2r(n, [REDFT01] * 2, lib=L, **kw)
        with pytest.raises(DspfftError):
            fwd.roundtrip(bad, got.ctypes.data)


# ---- lengths up to 32 in registers (TINY) and the guru-shaped batch interface (motion --blocksize 8x8x8) ----
@pytest.mark.parametrize("N", list(range(1, 33)))
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_tiny_lengths(N, dtype):
    f64 = dtype == "f64"
    for kind in (REDFT10, REDFT01):
        xb = ol.synth_f32(N + 50, N * 10).reshape(N, 10)
        xb = xb.astype(np.float64) * (1 + 2.0 ** -30) if f64 else xb
        ref = np.stack([ol.r2r_many(xb[:, t].astype(np.float64).copy(), [N], [kind]) for t in range(10)], axis=1)
        p = Plan.many_r2r([N], [kind], howmany=10, istride=10, idist=1, ostride=10, odist=1, lib=emul(), dtype=dtype)
        assert "TINY" in p.describe(), p.describe()
        got = run64(p, xb.copy()) if f64 else run(p, xb.copy())
        assert relerr(got, ref) < (1e-14 if f64 else 1e-6), (N, kind)
        # contiguous lines
        xc = np.ascontiguousarray(xb.T)
        p = Plan.many_r2r([N], [kind], howmany=10, idist=N, odist=N, lib=emul(), dtype=dtype)
        got = run64(p, xc.copy()) if f64 else run(p, xc.copy())
        assert relerr(got, ref.T) < (1e-14 if f64 else 1e-6), (N, kind)


@pytest.mark.parametrize("block", [(8, 8, 8), (4, 6, 16), (2, 3, 5)])
def test_guru_blocks_of_a_volume(block):
    """every block of a [D][H][W] volume in ONE plan (motion --blocksize WxHxD processes them one by one, motion.c:591-615)"""
    bd, bh, bw = block
    D, H, W = 2 * bd, 3 * bh, 2 * bw
    x = ol.synth_f32(77, D * H * W).reshape(D, H, W)
    dims = [(bd, H * W, H * W), (bh, W, W), (bw, 1, 1)]
    how = [(D // bd, bd * H * W, bd * H * W), (H // bh, bh * W, bh * W), (W // bw, bw, bw)]
    for kind in (REDFT10, REDFT01):
        p = Plan.guru(dims, how, [kind] * 3, lib=emul())
        assert p.describe().count("TINY") == 3, p.describe()
        got = run(p, x.copy())
        ref = np.empty((D, H, W))
        for z in range(0, D, bd):
            for y in range(0, H, bh):
                for xx in range(0, W, bw):
                    blk = np.ascontiguousarray(x[z:z + bd, y:y + bh, xx:xx + bw]).astype(np.float64)
                    ref[z:z + bd, y:y + bh, xx:xx + bw] = ol.r2r_many(blk.ravel(), [bd, bh, bw], [kind] * 3).reshape(bd, bh, bw)
        assert relerr(got, ref) < TOL, (block, kind)
    # the same geometry with a long axis: 1x1xD blocks = a temporal transform per pixel (motion --blocksize 1x1x0)
    p = Plan.guru([(D, H * W, H * W)], [(H * W, 1, 1)], [REDFT10], lib=emul())
    got = run(p, x.copy())
    ref = np.stack([ol.r2r_many(x.reshape(D, -1)[:, i].astype(np.float64).copy(), [D], [REDFT10]) for i in range(H * W)], axis=1).reshape(D, H, W)
    assert relerr(got, ref) < TOL


def test_guru_rejects_bad_arguments_and_tiny_fused_scan():
    with pytest.raises(DspfftError):
        Plan.guru([(8, 1, 1)] * 4, [], [REDFT10] * 4, lib=emul())
    with pytest.raises(DspfftError):
        Plan.guru([(8, 0, 1)], [], [REDFT10], lib=emul())
    # masked + accumulating execution through TINY passes (a 12x16 image: both axes in registers)
    h, w, c = 12, 16, 3
    L = emul()
    x = ol.synth_f32(8, h * w * c).reshape(h, w, c)
    coeffs = x.copy()
    Plan.image(h, w, c, REDFT10, lib=L).set_scale(1.0 / (4 * w * h)).execute(coeffs.ctypes.data)
    ids = np.zeros(h * w, dtype=np.uint32)
    step = (h * w + 3) // 4
    assert L.dspfft_scan_zigzag_frame_ids(ids.ctypes.data, w, h, step, None) == 0
    inv = Plan.image(h, w, c, REDFT01, lib=L)
    assert inv.describe().count("TINY") == 2
    acc = np.ascontiguousarray(np.broadcast_to(coeffs[0, 0], (h, w, c)).copy())
    work = np.zeros_like(acc)
    for f in range(4):
        inv.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc.ctypes.data, ids.ctypes.data, f, c)
    assert np.abs(acc - x).max() < 1e-5


@pytest.mark.parametrize("N,lines", [(8, 600), (5, 257), (32, 300), (16, 256)])
def test_tiny_packed_lines_many_chunks(N, lines):
    """contiguous lines back to back go through LDS in chunks of 256 lines; several chunks and a ragged tail"""
    x = ol.synth_f32(N * 3 + lines, N * lines).reshape(lines, N)
    for kind in (REDFT10, REDFT01):
        p = Plan.many_r2r([N], [kind], howmany=lines, idist=N, odist=N, lib=emul())
        assert "packed" in p.describe()
        ref = ol.r2r_many(x.astype(np.float64).ravel(), [N], [kind], howmany=lines, idist=N, odist=N).reshape(lines, N)
        assert relerr(run(p, x.copy()), ref) < 1e-6
    # two batch levels: 3 separate runs of `lines` packed lines each, with a gap between runs
    buf = ol.synth_f32(9, 3 * (N * lines + 7)).reshape(3, N * lines + 7)
    p = Plan.guru([(N, 1, 1)], [(lines, N, N), (3, N * lines + 7, N * lines + 7)], [REDFT10], lib=emul())
    assert "packed" in p.describe()
    got = buf.copy(); p.execute(got.ctypes.data)
    for r in range(3):
        ref = ol.r2r_many(buf[r, :N * lines].astype(np.float64).copy(), [N], [REDFT10], howmany=lines, idist=N, odist=N)
        assert relerr(got[r, :N * lines], ref) < 1e-6
        assert np.array_equal(got[r, N * lines:], buf[r, N * lines:])


# ---- randomised geometries: FFTW's advanced interface (rank, n, howmany, stride, dist, embed; in or out of place) ----
def _random_case(rng):
    rank = int(rng.integers(1, 4))
    n = [int(rng.choice([1, 2, 3, 4, 5, 6, 8, 9, 12, 16, 17, 20, 33, 34, 36, 37, 40, 48, 74])) for _ in range(rank)]
    while np.prod(n) > 40000:
        n[int(rng.integers(0, rank))] = int(rng.choice([2, 3, 4, 6, 8]))
    layout = rng.choice(["planar", "interleaved"])
    howmany = int(rng.choice([1, 2, 3, 4, 5]))
    embed = [v + int(rng.choice([0, 0, 1, 3])) for v in n]
    embed[0] = n[0]                                            # FFTW ignores embed[0] beyond sizing
    if layout == "interleaved":
        stride, dist = howmany, 1
        total = int(np.prod(embed)) * howmany
    else:
        stride = 1
        dist = int(np.prod(embed)) + int(rng.choice([0, 0, 5]))
        total = dist * howmany
    kinds = [int(rng.choice([REDFT10, REDFT01])) for _ in range(rank)]
    return rank, n, howmany, embed, stride, dist, total, kinds


@pytest.mark.parametrize("seed", range(48))
def test_random_advanced_interface_geometries(seed):
    rng = np.random.default_rng(1000 + seed)
    rank, n, howmany, embed, stride, dist, total, kinds = _random_case(rng)
    f64 = bool(seed % 3 == 0)
    oop = bool(rng.integers(0, 2))
    x = ol.synth_f32(seed + 1, total)
    x = x.astype(np.float64) if f64 else x
    ref_full = ol.r2r_many(x.astype(np.float64), n, kinds, howmany=howmany, inembed=embed, istride=stride, idist=dist,
                           onembed=embed, ostride=stride, odist=dist)
    p = Plan.many_r2r(n, kinds, howmany=howmany, inembed=embed, istride=stride, idist=dist, onembed=embed, ostride=stride, odist=dist,
                      lib=emul(), dtype="f64" if f64 else "f32")
    if oop:
        out = np.full(total, 7.0, dtype=x.dtype)
        src = x.copy()
        p.execute(src.ctypes.data, out.ctypes.data)
        assert np.array_equal(src, x), p.describe()
        got = out
        # positions the transform does not own keep the output buffer's previous content
        ref = np.full(total, 7.0)
    else:
        got = x.copy()
        p.execute(got.ctypes.data)
        ref = x.astype(np.float64).copy()
    # element offsets the plan owns
    idx = np.zeros(1, dtype=np.int64)
    mult = stride
    for a in range(rank - 1, -1, -1):
        idx = (idx[None, :] + (np.arange(n[a]) * mult)[:, None]).ravel()
        mult *= embed[a]
    idx = (idx[None, :] + (np.arange(howmany) * dist)[:, None]).ravel()
    ref[idx] = ref_full[idx]
    tol = 5e-13 if f64 else 3e-6
    scale = max(np.abs(ref_full[idx]).max(), 1e-30)
    assert np.abs(got.astype(np.float64) - ref).max() <= tol * scale, (n, howmany, embed, stride, dist, kinds, oop, p.describe())


def _oracle_along_axis(arr, axis, kind):
    """the definition along one axis of a dense float64 array"""
    moved = np.ascontiguousarray(np.moveaxis(arr, axis, -1))
    n = moved.shape[-1]
    lines = moved.reshape(-1, n)
    out = ol.r2r_many(lines.ravel(), [n], [kind], howmany=lines.shape[0], idist=n, odist=n).reshape(moved.shape)
    return np.moveaxis(out, -1, axis)


@pytest.mark.parametrize("seed", range(24))
def test_random_guru_geometries(seed):
    """guru-shaped plans on a dense array: a random subset of 1..3 axes is transformed, the rest are batch dimensions"""
    rng = np.random.default_rng(9000 + seed)
    nd = int(rng.integers(2, 6))
    shape = [int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 16, 20, 33, 37])) for _ in range(nd)]
    while np.prod(shape) > 30000:
        shape[int(rng.integers(0, nd))] = int(rng.choice([2, 3, 4]))
    strides = [int(np.prod(shape[i + 1:])) for i in range(nd)]
    rank = int(rng.integers(1, min(3, nd) + 1))
    taxes = sorted(rng.choice(nd, size=rank, replace=False).tolist())
    kinds = [int(rng.choice([REDFT10, REDFT01])) for _ in range(rank)]
    dims = [(shape[a], strides[a], strides[a]) for a in taxes]
    how = [(shape[a], strides[a], strides[a]) for a in range(nd) if a not in taxes]
    rng.shuffle(how)
    f64 = bool(seed % 2)
    x = ol.synth_f32(seed + 3, int(np.prod(shape))).reshape(shape)
    x = x.astype(np.float64) if f64 else x
    p = Plan.guru(dims, how, kinds, lib=emul(), dtype="f64" if f64 else "f32")
    got = x.copy()
    p.execute(got.ctypes.data)
    ref = x.astype(np.float64)
    for a, k in zip(taxes, kinds):
        ref = _oracle_along_axis(ref, a, k)
    assert relerr(got, ref) < (5e-13 if f64 else 3e-6), (shape, taxes, kinds, p.describe())


@pytest.mark.parametrize("lpw", [2, 4, 16])
def test_generic_row_pass_with_several_lines_per_workgroup(lpw):
    """short lines share a workgroup (the planner does this for big batches of short lines; forced here), with a ragged last group"""
    os.environ["DSPFFT_ROW_LPW"] = str(lpw)
    os.environ["DSPFFT_NO_TINY"] = "1"
    try:
        for (h, w, c) in [(37, 64, 3), (5, 36, 1), (21, 40, 2)]:
            x = ol.synth_f32(h + w, h * w * c).reshape(h, w, c)
            for kind in (REDFT10, REDFT01):
                p = Plan.image(h, w, c, kind, lib=emul())
                assert f"x{lpw}" in p.describe().splitlines()[1], p.describe()
                ref = ol.dct2d_interleaved(x.astype(np.float64), kind, impl="port")
                assert relerr(run(p, x.copy()), ref) < TOL, (h, w, c, kind, p.describe())
                p64 = Plan.image(h, w, c, kind, lib=emul(), dtype="f64")
                assert relerr(run64(p64, x.astype(np.float64)), ref) < 5e-13
        # fused scan step through it
        h, w, c = 37, 64, 3
        L = emul()
        x = ol.synth_f32(2, h * w * c).reshape(h, w, c)
        coeffs = x.copy()
        Plan.image(h, w, c, REDFT10, lib=L).set_scale(1.0 / (4 * w * h)).execute(coeffs.ctypes.data)
        ids = np.zeros(h * w, dtype=np.uint32)
        assert L.dspfft_scan_zigzag_frame_ids(ids.ctypes.data, w, h, (h * w + 2) // 3, None) == 0
        inv = Plan.image(h, w, c, REDFT01, lib=L)
        acc = np.ascontiguousarray(np.broadcast_to(coeffs[0, 0], (h, w, c)).copy())
        work = np.zeros_like(acc)
        for f in range(3):
            inv.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc.ctypes.data, ids.ctypes.data, f, c)
        assert np.abs(acc - x).max() < 1e-5
    finally:
        del os.environ["DSPFFT_ROW_LPW"], os.environ["DSPFFT_NO_TINY"]


# ---- double-precision specialised kernels (spec_list.h DSPFFT_*_SPECS_F64: RowSpecT<double> / ColSpecT<double>) ----
@pytest.mark.parametrize("h,w,c", [(512, 512, 3), (540, 960, 3), (720, 1280, 3), (1024, 1024, 3), (720, 1280, 1)])
@pytest.mark.parametrize("kind", [REDFT10, REDFT01])
def test_f64_specialised_sizes(h, w, c, kind):
    x = ol.synth_f32(h + w, h * w * c).astype(np.float64).reshape(h, w, c) * (1 + 2.0 ** -30)
    p = Plan.image(h, w, c, kind, lib=emul(), dtype="f64")
    d = p.describe()
    assert "ROW* f64" in d and "COL* f64" in d, d
    ref = ol.dct2d_interleaved(x, kind, impl="port", threads=8)
    assert relerr(run64(p, x.copy()), ref) < 1e-13, d
    # out of place, and the generic kernels give the same answer to rounding
    out = np.empty_like(x)
    assert relerr(run64(p, x.copy(), out), ref) < 1e-13
    os.environ["DSPFFT_NO_SPEC"] = "1"
    try:
        pg = Plan.image(h, w, c, kind, lib=emul(), dtype="f64")
    finally:
        del os.environ["DSPFFT_NO_SPEC"]
    assert "*" not in pg.describe()
    assert relerr(run64(pg, x.copy()), ref) < 1e-13


def test_f64_specialised_fused_scan_step():
    """masked loads + accumulate stores in the double kernels (scan.c:421-459 with COEFF_PRECISION=D)"""
    h, w, c = 512, 512, 3
    L = emul()
    x = ol.synth_f32(77, h * w * c).astype(np.float64).reshape(h, w, c)
    coeffs = x.copy()
    Plan.image(h, w, c, REDFT10, lib=L, dtype="f64").set_scale(1.0 / (4 * w * h)).execute(coeffs.ctypes.data)
    inv = Plan.image(h, w, c, REDFT01, lib=L, dtype="f64")
    assert "ROW* f64" in inv.describe() and "COL* f64" in inv.describe()
    ids = np.zeros(h * w, dtype=np.uint32)
    nframes = 3
    assert L.dspfft_scan_zigzag_frame_ids(ids.ctypes.data, w, h, (h * w + nframes - 1) // nframes, None) == 0
    acc = np.ascontiguousarray(np.broadcast_to(coeffs[0, 0], (h, w, c)).copy())
    work = np.zeros_like(acc)
    for f in range(nframes):
        inv.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc.ctypes.data, ids.ctypes.data, f, c)
    assert np.abs(acc - x).max() < 1e-12


# ---- small blocks: all axes in one pass (block_core.h) ----
@pytest.mark.parametrize("block,vol", [((8, 8, 8), (16, 16, 8 * 37)), ((4, 4, 4), (8, 12, 4 * 70)), ((16, 16, 16), (16, 32, 16 * 5)), ((1, 8, 8), (3, 16, 8 * 33)),
                                       ((1, 16, 16), (2, 16, 16 * 3)), ((8, 8, 16), (8, 16, 64)), ((4, 8, 8), (8, 8, 8 * 40)), ((8, 12, 8), (8, 24, 32))])
def test_fused_block_pass(block, vol, monkeypatch):
    """WxHxD blocks of a [D][H][W] volume (motion --blocksize): the one-pass kernel against the per-axis TINY passes and the oracle"""
    bd, bh, bw = block
    D, H, W = vol
    x = ol.synth_f32(D + H + W, D * H * W).reshape(D, H, W)
    dims = [(bd, H * W, H * W), (bh, W, W), (bw, 1, 1)]
    how = [(D // bd, bd * H * W, bd * H * W), (H // bh, bh * W, bh * W), (W // bw, bw, bw)]
    dims = [d for d in dims if d[0] > 1]
    fusable = (bw, bh, bd) != (8, 12, 8)
    for kind in (REDFT10, REDFT01):
        p = Plan.guru(dims, how, [kind] * len(dims), lib=emul()).set_scale(0.37)
        for a in range(len(dims)):
            p.set_axis_scale0(a, 1.25 + a, 0.5 + 0.25 * a)
        assert ("BLOCK" in p.describe()) == fusable, p.describe()
        monkeypatch.setenv("DSPFFT_NO_BLOCK", "1")
        q = Plan.guru(dims, how, [kind] * len(dims), lib=emul()).set_scale(0.37)
        monkeypatch.delenv("DSPFFT_NO_BLOCK")
        for a in range(len(dims)):
            q.set_axis_scale0(a, 1.25 + a, 0.5 + 0.25 * a)
        assert "BLOCK" not in q.describe()
        got, ref = run(p, x.copy()), run(q, x.copy())
        assert relerr(got, ref.astype(np.float64)) < 1e-6, (block, kind)
        out = np.empty_like(x)
        assert np.array_equal(run(p, x.copy(), out), got)               # out of place
        # an unaligned buffer takes the per-axis passes
        raw = np.empty(D * H * W + 1, dtype=np.float32)
        u = raw[1:] if raw.ctypes.data % 16 == 0 else raw[:-1]
        u[...] = x.ravel()
        p.execute(u.ctypes.data)
        assert relerr(u.reshape(D, H, W), ref.astype(np.float64)) < 1e-6
    # one block against the definition
    p = Plan.guru(dims, how, [REDFT10] * len(dims), lib=emul())
    got = run(p, x.copy())
    blk = np.ascontiguousarray(x[:bd, :bh, :bw]).astype(np.float64)
    ref = ol.r2r_many(blk.ravel(), [v for v in (bd, bh, bw) if v > 1], [REDFT10] * len(dims)).reshape(bd, bh, bw)
    assert relerr(got[:bd, :bh, :bw], ref) < TOL


def _aligned_f32(n):
    raw = np.empty(n * 4 + 64, dtype=np.uint8)
    o = (-raw.ctypes.data) % 64
    return raw[o:o + 4 * n].view(np.float32)


@pytest.mark.parametrize("block,nblocks", [((8, 8, 8), 37), ((4, 4, 4), 130), ((1, 8, 8), 70), ((16, 16, 16), 3), ((8, 16, 4), 9)])
@pytest.mark.parametrize("what", ["plain", "filter", "quant_u8"])
def test_fused_block_roundtrip_block_major(block, nblocks, what, monkeypatch):
    """motion's per-block pipeline on a block-major stack (the reference's per-block buffers, motion.c:591-776): one pass against the
    unfused passes (DSPFFT_NO_BLOCK=1), 8-bit ends included"""
    import ctypes as C
    L = emul()
    bd, bh, bw = block
    vol = bd * bh * bw
    n = [v for v in (bd, bh, bw) if v > 1]
    rank = len(n)
    r2 = float(np.sqrt(2.0))
    nrm = 1.0 / np.prod([2.0 * v for v in n])
    u8 = ol.synth_u8(nblocks + vol, nblocks * vol)
    x = _aligned_f32(nblocks * vol); x[...] = u8.astype(np.float32)

    def plans():
        fwd = Plan.many_r2r(n, [REDFT10] * rank, howmany=nblocks, idist=vol, odist=vol, lib=L).set_scale(2 * r2)
        inv = Plan.many_r2r(n, [REDFT01] * rank, howmany=nblocks, idist=vol, odist=vol, lib=L, first_axis_first=True).set_scale(nrm / (2 * r2))
        for a in range(rank):
            fwd.set_axis_scale0(a, 1.0, 1.0 / r2); inv.set_axis_scale0(a, r2, 1.0)
        return fwd, inv
    flt = None
    if what == "filter":
        flt = dict(active=(bd, bh, bw), minbuf_hw=(bh, bw), block_depth=bd, band_begin=(0, 1, 1), band_end=(bd, bh - 1, bw), damp=0.25, boost=1.5, preserve_dc=1)
    elif what == "quant_u8":
        flt = dict(active=(bd, bh, bw), minbuf_hw=(bh, bw), block_depth=bd, band_begin=(0, 0, 0), band_end=(bd, bh, bw), quantizer=6.0)
    res = {}
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("DSPFFT_NO_BLOCK", raising=False)
        else:
            monkeypatch.setenv("DSPFFT_NO_BLOCK", "1")
        fwd, inv = plans()
        assert ("BLOCK" in fwd.describe()) == fused and ("block-major" in fwd.describe()) == fused, fwd.describe()
        coded = np.zeros(1, dtype=np.uint64)
        if what == "quant_u8":
            inb = np.ascontiguousarray(u8); outb = np.zeros_like(u8); work = _aligned_f32(nblocks * vol)
            fwd.roundtrip_u8(inv, inb.ctypes.data, outb.ctypes.data, work.ctypes.data, 1.0, filter=flt, d_coded=coded.ctypes.data)
            res[fused] = (outb.copy(), int(coded[0]))
        else:
            buf = _aligned_f32(nblocks * vol); buf[...] = x
            fwd.roundtrip(inv, buf.ctypes.data, filter=flt, d_coded=coded.ctypes.data if flt else 0)
            res[fused] = (buf.copy(), int(coded[0]))
    if what == "quant_u8":
        assert np.abs(res[True][0].astype(np.int32) - res[False][0].astype(np.int32)).max() <= 1     # a value on a rounding edge may flip
        assert (res[True][0] != res[False][0]).mean() < 1e-3
        assert abs(res[True][1] - res[False][1]) <= max(4, res[False][1] // 10000) and res[False][1] > 0
    else:
        assert np.abs(res[True][0] - res[False][0]).max() < 2e-4 * 255
        if what == "plain":
            assert np.abs(res[True][0] - x).max() < 1e-3


@pytest.mark.parametrize("block", [(8, 8, 8), (4, 16, 8), (1, 8, 8)])
def test_fused_block_roundtrip_volume_layout(block):
    """the blocks of a [D][H][W] volume where they lie (guru plans): the one-pass pipeline equals the block-major one on the rearranged data"""
    L = emul()
    bd, bh, bw = block
    D, H, W = 2 * bd, 2 * bh, 9 * bw
    nb = (D // bd) * (H // bh) * (W // bw)
    vol = bd * bh * bw
    n = [v for v in (bd, bh, bw) if v > 1]
    rank = len(n)
    u8 = ol.synth_u8(5, D * H * W).reshape(D, H, W)
    dims = [d for d in [(bd, H * W, H * W), (bh, W, W), (bw, 1, 1)] if d[0] > 1]
    how = [(D // bd, bd * H * W, bd * H * W), (H // bh, bh * W, bh * W), (W // bw, bw, bw)]
    flt = dict(active=(bd, bh, bw), minbuf_hw=(bh, bw), block_depth=bd, band_begin=(0, 1, 0), band_end=(bd, bh, bw - 1), damp=0.5, boost=1.25, preserve_dc=1, quantizer=3.0)
    nrm = 1.0 / np.prod([2.0 * v for v in n])
    # volume layout
    fv = Plan.guru(dims, how, [REDFT10] * rank, lib=L)
    iv = Plan.guru(dims, how, [REDFT01] * rank, lib=L).set_scale(nrm)
    assert "side by side" in fv.describe()
    cv = np.zeros(1, dtype=np.uint64)
    outv = np.zeros_like(u8); work = _aligned_f32(D * H * W)
    fv.roundtrip_u8(iv, np.ascontiguousarray(u8).ctypes.data, outv.ctypes.data, work.ctypes.data, 1.0, filter=flt, d_coded=cv.ctypes.data)
    # block-major layout of the same blocks
    bm = np.ascontiguousarray(u8.reshape(D // bd, bd, H // bh, bh, W // bw, bw).transpose(0, 2, 4, 1, 3, 5))
    fb = Plan.many_r2r(n, [REDFT10] * rank, howmany=nb, idist=vol, odist=vol, lib=L)
    ib = Plan.many_r2r(n, [REDFT01] * rank, howmany=nb, idist=vol, odist=vol, lib=L, first_axis_first=True).set_scale(nrm)
    cb = np.zeros(1, dtype=np.uint64)
    outb = np.zeros_like(bm); workb = _aligned_f32(D * H * W)
    fb.roundtrip_u8(ib, bm.ctypes.data, outb.ctypes.data, workb.ctypes.data, 1.0, filter=flt, d_coded=cb.ctypes.data)
    back = outb.reshape(D // bd, H // bh, W // bw, bd, bh, bw).transpose(0, 3, 1, 4, 2, 5).reshape(D, H, W)
    assert np.array_equal(outv, back) and int(cv[0]) == int(cb[0]) > 0
    # float in place, and an unaligned buffer with a filter is refused rather than filtered at the wrong positions
    buf = _aligned_f32(D * H * W); buf[...] = u8.ravel()
    fv.roundtrip(iv, buf.ctypes.data, filter=flt)
    assert np.abs(np.clip(np.floor(buf + 0.5), 0, 255).reshape(D, H, W) - outv).max() <= 1
    raw = np.zeros(D * H * W + 1, dtype=np.float32)
    u = raw[1:] if raw.ctypes.data % 16 == 0 else raw[:-1]
    with pytest.raises(DspfftError):
        fv.roundtrip(iv, u.ctypes.data, filter=flt)


def test_dense_lines_staged_in_memory(monkeypatch):
    """a line too long for LDS is staged in a device array of the plan (engine.cpp DENSE: the reference never checks for a NULL plan,
    spec.c:63-64); forced here on small prime lengths, in place, with a batch and with the fused scales"""
    monkeypatch.setenv("DSPFFT_DENSE_STAGED", "1")
    monkeypatch.setenv("DSPFFT_NO_BLUESTEIN", "1")
    for (h, w, c) in ((37, 5, 3), (3, 41, 2)):
        x = ol.synth_f32(h * 100 + w, h * w * c).reshape(h, w, c)
        for kind in (REDFT10, REDFT01):
            p = Plan.image(h, w, c, kind, lib=emul())
            assert "staged in device memory" in p.describe(), p.describe()
            ref = ol.dct2d_interleaved(x.astype(np.float64), kind, impl="port")
            assert relerr(run(p, x.copy()), ref) < TOL
    p = Plan.many_r2r([43], [REDFT10], lib=emul()).set_scale(0.25).set_axis_scale0(0, 3.0, 0.5)
    x = ol.synth_f32(9, 43)
    want = ol.r2r_many(x.astype(np.float64) * np.where(np.arange(43) == 0, 3.0, 1.0), [43], [ol.REDFT10]) * 0.25
    want[0] *= 0.5
    assert relerr(run(p, x.copy()), want) < TOL


def test_input_window_is_declined_where_it_is_not_implemented():
    """dspfft_plan_set_input_window on the emulation backend (no listed specialised kernels in play for this shape): 0 = not honoured,
    the zeros must really be stored; bad arguments are errors"""
    p = Plan.many_r2r([37], [REDFT01], howmany=6, istride=6, idist=1, ostride=6, odist=1, lib=emul())
    assert p.set_input_window(0, 0, 10) is False
    with pytest.raises(DspfftError):
        p.set_input_window(1, 0, 10)
    with pytest.raises(DspfftError):
        p.set_input_window(0, 10, 5)


# ---- channel lines (dct_spec.h RowChanSpecT): interleaved double lines of 3840 / 4096 pixels run one workgroup per (line, channel) ----
@pytest.mark.parametrize("w", [3840, 4096])
@pytest.mark.parametrize("h", [10, 3])          # 10: one full group of eight lines + the tail of chan_work; 3: tail only
def test_f64_channel_lines_match_the_interleaved_kernel_and_the_port(w, h, monkeypatch):
    c = 3
    L = emul()
    x = ol.synth_f32(w + h, h * w * c).astype(np.float64).reshape(h, w, c) + 1e-9 * np.arange(h * w * c).reshape(h, w, c)
    for kind in (REDFT10, REDFT01):
        p = Plan.image(h, w, c, kind, lib=L, dtype="f64").set_scale(0.37).set_axis_scale0(1, 0.5, 0.7)
        assert f"ROW* f64 N={w} C=3" in p.describe() and "as 3 channel lines" in p.describe(), p.describe()
        ref = ol.dct2d_interleaved(x, kind, impl="port")
        monkeypatch.setenv("DSPFFT_ROW_CHAN", "1")
        got = run64(p, x.copy())
        out = np.zeros_like(x)
        run64(p, x.copy(), out)                               # out of place
        monkeypatch.setenv("DSPFFT_ROW_CHAN", "0")
        base = run64(p, x.copy())
        assert np.array_equal(got, base) and np.array_equal(out, base)      # same butterflies, same order: bit-identical
        q = Plan.image(h, w, c, kind, lib=L, dtype="f64").set_scale(0.37)        # no per-index scales: the port's transform times 0.37
        monkeypatch.setenv("DSPFFT_ROW_CHAN", "1")
        assert relerr(run64(q, x.copy()), 0.37 * ref) < TOL64
    # the fused scan step (owner-id mask on the first pass, accumulation in the row pass's stores) through channel lines
    coeffs = x.copy()
    Plan.image(h, w, c, REDFT10, lib=L, dtype="f64").set_scale(1.0 / (4 * w * h)).execute(coeffs.ctypes.data)
    ids = np.zeros(h * w, dtype=np.uint32)
    assert L.dspfft_scan_zigzag_frame_ids(ids.ctypes.data, w, h, (h * w + 2) // 3, None) == 0
    inv = Plan.image(h, w, c, REDFT01, lib=L, dtype="f64")
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("DSPFFT_ROW_CHAN", on)
        acc = np.ascontiguousarray(np.broadcast_to(coeffs[0, 0], (h, w, c)).copy())
        work = np.zeros_like(acc)
        for f in range(3):
            inv.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc.ctypes.data, ids.ctypes.data, f, c)
        res[on] = acc
    assert np.array_equal(res["1"], res["0"])
    assert np.abs(res["1"] - x).max() < 1e-13


# ---- input window / alternating output on a listed ROW REDFT01 pass (zoom's x stage when it runs last: zoom_fft.hip) ----
def test_row_pass_honours_window_and_alternate_with_compact_input_lines():
    """samples outside the window are not read, so a line needs to hold only its window: the cosine part's lines are cw pixels long, the sine
    part's cw - 1 pixels and addressed from `lo` pixels before their start; the sine part's store alternates its sign and accumulates"""
    L = emul()
    N, c, lines, cw = 640, 3, 5, 160
    full = np.zeros((lines, N, c))
    full[:, :cw] = ol.synth_f32(3, lines * cw * c).reshape(lines, cw, c) - 0.5
    want = np.stack([ol.r2r_many(full[j], [N], [ol.REDFT01], howmany=c, istride=c, idist=1, ostride=c, odist=1, impl="port").reshape(N, c) for j in range(lines)])
    pa = Plan.guru([(N, c, c)], [(c, 1, 1), (lines, cw * c, N * c)], [REDFT01], lib=L).set_scale(0.5)
    assert "ROW*" in pa.describe(), pa.describe()
    assert pa.set_input_window(0, 0, cw) is True
    compact = np.ascontiguousarray(full[:, :cw], dtype=np.float32)
    out = np.full((lines, N, c), np.nan, dtype=np.float32)
    pa.execute(compact.ctypes.data, out.ctypes.data)
    assert np.abs(out - 0.5 * want).max() <= 2e-6 * np.abs(want).max()
    # the sine part: window [lo, N), lo = N - cw + 1; compact lines of cw - 1 pixels addressed from lo pixels before their start
    lo = N - cw + 1
    fe = np.zeros((lines, N, c))
    fe[:, lo:] = ol.synth_f32(4, lines * (cw - 1) * c).reshape(lines, cw - 1, c) - 0.5
    we = np.stack([ol.r2r_many(fe[j], [N], [ol.REDFT01], howmany=c, istride=c, idist=1, ostride=c, odist=1, impl="port").reshape(N, c) for j in range(lines)])
    pe = Plan.guru([(N, c, c)], [(c, 1, 1), (lines, (cw - 1) * c, N * c)], [REDFT01], lib=L).set_scale(-0.25)
    assert pe.set_input_window(0, lo, N) is True and pe.set_output_alternate(0) is True
    ce = np.ascontiguousarray(fe[:, lo:], dtype=np.float32)
    guard = np.concatenate([np.full(8, np.nan, dtype=np.float32), ce.ravel(), np.full(8, np.nan, dtype=np.float32)])     # NaNs either side: nothing else is read
    acc = out.copy()
    pe.execute_masked_accumulate(guard[8:].ctypes.data - lo * c * 4, acc.ctypes.data, acc.ctypes.data)
    sign = np.where(np.arange(N) % 2 == 1, -1.0, 1.0)[None, :, None]
    assert np.abs(acc - (0.5 * want - 0.25 * sign * we)).max() <= 2e-6 * (np.abs(want).max() + np.abs(we).max())
    # the two parts in ONE launch (dspfft_execute_sum2: the cosine part's line waits in registers, the frame is written once)
    both = np.full((lines, N, c), np.nan, dtype=np.float32)
    pa.execute_sum2(pe, compact.ctypes.data, guard[8:].ctypes.data - lo * c * 4, both.ctypes.data)
    assert np.abs(both - acc).max() <= 1e-6 * (np.abs(want).max() + np.abs(we).max())
    # ... and where the two plans do not share a row kernel: one execution after the other, same result
    pc = Plan.guru([(N, c * lines, c * lines)], [(c * lines, 1, 1)], [REDFT01], lib=L).set_scale(0.5)        # a column plan over the transposed data
    xt = np.ascontiguousarray(full.transpose(1, 0, 2), dtype=np.float32)
    fb = np.full((N, lines, c), np.nan, dtype=np.float32)
    pc.execute_sum2(pc, xt.ctypes.data, xt.ctypes.data, fb.ctypes.data)
    assert np.abs(fb.transpose(1, 0, 2) - want).max() <= 2e-6 * np.abs(want).max()
    # input modulation: both parts read ONE array T through multiplier tables, the second mirrored about N (zoom's x stage)
    T = np.ascontiguousarray(ol.synth_f32(6, lines * cw * c).reshape(lines, cw, c) - 0.5, dtype=np.float32)
    ma, mb = ol.synth_f32(7, cw).astype(np.float32), ol.synth_f32(8, cw).astype(np.float32)
    fa = np.zeros((lines, N, c)); fa[:, :cw] = T * ma[None, :, None]
    fb = np.zeros((lines, N, c)); fb[:, lo:] = (T * mb[None, :, None])[:, N - np.arange(lo, N)]          # sample x <- position N - x
    wa = np.stack([ol.r2r_many(fa[j], [N], [ol.REDFT01], howmany=c, istride=c, idist=1, ostride=c, odist=1, impl="port").reshape(N, c) for j in range(lines)])
    wb = np.stack([ol.r2r_many(fb[j], [N], [ol.REDFT01], howmany=c, istride=c, idist=1, ostride=c, odist=1, impl="port").reshape(N, c) for j in range(lines)])
    qa = Plan.guru([(N, c, c)], [(c, 1, 1), (lines, cw * c, N * c)], [REDFT01], lib=L).set_scale(0.5)
    qb = Plan.guru([(N, c, c)], [(c, 1, 1), (lines, cw * c, N * c)], [REDFT01], lib=L).set_scale(-0.25)
    assert qa.set_input_modulation(0, ma.ctypes.data) is False                      # no window yet: declined
    assert qa.set_input_window(0, 0, cw) is True and qa.set_input_modulation(0, ma.ctypes.data) is True
    assert qb.set_input_window(0, lo, N) is True and qb.set_output_alternate(0) is True and qb.set_input_modulation(0, mb.ctypes.data, N) is True
    both = np.full((lines, N, c), np.nan, dtype=np.float32)
    qa.execute_sum2(qb, T.ctypes.data, T.ctypes.data, both.ctypes.data)
    assert np.abs(both - (0.5 * wa - 0.25 * sign * wb)).max() <= 2e-6 * (np.abs(wa).max() + np.abs(wb).max())
    with pytest.raises(DspfftError):
        qa.set_input_modulation(0, ma.ctypes.data, cw - 5)                             # the window reaches beyond the reversal point
    assert qa.set_input_window(0, 0, cw) is True                                      # setting a window again drops the modulation
    alone = np.full((lines, N, c), np.nan, dtype=np.float32)
    qa.execute(T.ctypes.data, alone.ctypes.data)
    fz = np.zeros((lines, N, c)); fz[:, :cw] = T
    wz = np.stack([ol.r2r_many(fz[j], [N], [ol.REDFT01], howmany=c, istride=c, idist=1, ostride=c, odist=1, impl="port").reshape(N, c) for j in range(lines)])
    assert np.abs(alone - 0.5 * wz).max() <= 2e-6 * np.abs(wz).max()
    two = Plan.image(8, 16, 3, REDFT01, lib=L)                # two passes: cannot be the accumulating second plan
    z = np.zeros((8, 16, 3), dtype=np.float32)
    with pytest.raises(DspfftError, match="one-pass"):
        two.execute_sum2(two, z.ctypes.data, z.ctypes.data, np.zeros_like(z).ctypes.data)
    # turned off again / a forward plan / a double plan: not honoured
    assert pa.set_input_window(0, 0, 0) is False
    assert Plan.guru([(N, c, c)], [(c, 1, 1), (lines, N * c, N * c)], [REDFT10], lib=L).set_input_window(0, 0, cw) is False
    assert Plan.guru([(N, c, c)], [(c, 1, 1), (lines, N * c, N * c)], [REDFT01], lib=L, dtype="f64").set_output_alternate(0) is False


@pytest.mark.parametrize("seed", range(6))
def test_row_window_modulation_sum2_random_geometry(seed):
    """random listed row lengths, windows, line counts and pitches: the sum of a windowed modulated part and a mirrored, alternating one in
    one launch equals the zero-padded transforms of the port (edge cases: one-sample windows, windows touching either end, one line)"""
    rng = np.random.default_rng(1000 + seed)
    L = emul()
    N = int(rng.choice([256, 512, 640, 720, 800, 960]))
    c, lines = 3, int(rng.integers(1, 5))
    cw = int(rng.choice([1, 2, N // 4, N // 2, N - 1]))
    hi_a = cw
    lo_b = N - cw + 1 if cw > 1 else N - 1           # the mirrored part's window (lo_b, N) maps to positions [1, N - lo_b]
    pitch = max(cw, 2) * c + int(rng.integers(0, 3)) * c      # input lines may be further apart than their window (the mirrored part reads position 1 at least)
    T = np.ascontiguousarray(rng.standard_normal((lines, pitch // c, c)).astype(np.float32))
    ma, mb = rng.standard_normal(pitch // c).astype(np.float32), rng.standard_normal(pitch // c).astype(np.float32)
    fa = np.zeros((lines, N, c)); fa[:, :hi_a] = (T * ma[None, :, None])[:, :hi_a]
    fb = np.zeros((lines, N, c))
    xs = np.arange(lo_b, N)
    fb[:, xs] = (T * mb[None, :, None])[:, N - xs]
    tr = lambda f: np.stack([ol.r2r_many(f[j], [N], [ol.REDFT01], howmany=c, istride=c, idist=1, ostride=c, odist=1, impl="port").reshape(N, c) for j in range(lines)])
    wa, wb = tr(fa), tr(fb)
    sign = np.where(np.arange(N) % 2 == 1, -1.0, 1.0)[None, :, None]
    qa = Plan.guru([(N, c, c)], [(c, 1, 1), (lines, pitch, N * c)], [REDFT01], lib=L).set_scale(0.5)
    qb = Plan.guru([(N, c, c)], [(c, 1, 1), (lines, pitch, N * c)], [REDFT01], lib=L).set_scale(-0.25)
    assert "ROW*" in qa.describe(), qa.describe()
    assert qa.set_input_window(0, 0, hi_a) and qa.set_input_modulation(0, ma.ctypes.data)
    assert qb.set_input_window(0, lo_b, N) and qb.set_output_alternate(0) and qb.set_input_modulation(0, mb.ctypes.data, N)
    both = np.full((lines, N, c), np.nan, dtype=np.float32)
    qa.execute_sum2(qb, T.ctypes.data, T.ctypes.data, both.ctypes.data)
    want = 0.5 * wa - 0.25 * sign * wb
    assert np.abs(both - want).max() <= 3e-6 * max(1e-3, np.abs(wa).max() + np.abs(wb).max()), (N, cw, lines, pitch)


def test_column_pass_honours_modulation_and_mirror():
    """dspfft_plan_set_input_modulation on a listed COL REDFT01 pass: rows of the window are read from (mirrored) rows of a SHORTER array
    with another row pitch and multiplied by a per-row table on the way in (zoom's y stage reads the coefficients themselves this way)"""
    L = emul()
    N, inner, ch, pitch = 1080, 64, 270, 96
    C0 = np.ascontiguousarray(ol.synth_f32(21, ch * pitch).reshape(ch, pitch) - 0.5, dtype=np.float32)
    ma, mb = ol.synth_f32(22, ch).astype(np.float32), ol.synth_f32(23, ch).astype(np.float32)
    fa = np.zeros((N, inner)); fa[:ch] = C0[:, :inner] * ma[:, None]
    lo = N - ch + 1
    fb = np.zeros((N, inner)); ys = np.arange(lo, N); fb[ys] = (C0[:, :inner] * mb[:, None])[N - ys]
    tr = lambda f: ol.r2r_many(f, [N], [ol.REDFT01], howmany=inner, istride=inner, idist=1, ostride=inner, odist=1, impl="port").reshape(N, inner)
    wa, wb = tr(fa), tr(fb)
    pa = Plan.guru([(N, pitch, inner)], [(inner, 1, 1)], [REDFT01], lib=L).set_scale(0.5)
    pb = Plan.guru([(N, pitch, inner)], [(inner, 1, 1)], [REDFT01], lib=L).set_scale(-0.5)
    if "COL*" not in pa.describe():
        pytest.skip("no listed column kernel for this shape in this build")
    assert pa.set_input_window(0, 0, ch) and pa.set_input_modulation(0, ma.ctypes.data)
    assert pb.set_input_window(0, lo, N) and pb.set_input_modulation(0, mb.ctypes.data, N) and pb.set_output_alternate(0)
    out = np.full((N, inner), np.nan, dtype=np.float32)
    pa.execute(C0.ctypes.data, out.ctypes.data)
    work = np.zeros((N, inner), dtype=np.float32)
    pb.execute_masked_accumulate(C0.ctypes.data, work.ctypes.data, out.ctypes.data)
    sign = np.where(np.arange(N) % 2 == 1, -1.0, 1.0)[:, None]
    want = 0.5 * wa - 0.5 * sign * wb
    assert np.abs(out - want).max() <= 3e-6 * (np.abs(wa).max() + np.abs(wb).max())
