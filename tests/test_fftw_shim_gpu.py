"""GPU (-m gpu): the FFTW-named host-pointer boundary (include/fftw3.h) and the plain-C harnesses
built on it (host/), against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fftw():
    import torch
    assert torch.cuda.is_available()
    from dspfun_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    ip = C.POINTER(C.c_int)
    lib.fftwf_alloc_real.restype = C.c_void_p
    lib.fftwf_alloc_real.argtypes = [C.c_size_t]
    lib.fftwf_free.argtypes = [C.c_void_p]
    lib.fftwf_plan_many_r2r.restype = C.c_void_p
    lib.fftwf_plan_many_r2r.argtypes = [C.c_int, ip, C.c_int, C.c_void_p, ip, C.c_int, C.c_int, C.c_void_p, ip, C.c_int, C.c_int, ip, C.c_uint]
    lib.fftwf_execute.argtypes = [C.c_void_p]
    lib.fftwf_destroy_plan.argtypes = [C.c_void_p]
    lib.fftw_alloc_real.restype = C.c_void_p
    lib.fftw_alloc_real.argtypes = [C.c_size_t]
    lib.fftw_free.argtypes = [C.c_void_p]
    lib.fftw_plan_many_r2r.restype = C.c_void_p
    lib.fftw_plan_many_r2r.argtypes = lib.fftwf_plan_many_r2r.argtypes
    lib.fftw_execute.argtypes = [C.c_void_p]
    lib.fftw_destroy_plan.argtypes = [C.c_void_p]
    return lib


def _ia(v):
    return (C.c_int * len(v))(*v)


def _host_array(ptr, n, dtype):
    ct = C.c_float if dtype == np.float32 else C.c_double
    return np.ctypeslib.as_array((ct * n).from_address(ptr))


def test_spec_plan_in_place_on_host_pointers(fftw):
    # spec/spec.c:59-64 exactly: alloc_real, plan_many_r2r(2,{h,w},d,f,NULL,d,1,f,NULL,d,1,REDFT10^2,ESTIMATE), execute
    h, w, d = 256, 256, 3
    x = (ol.synth_u8(0xD5F0001, h * w * d).astype(np.float32) / np.float32(255)).reshape(h, w, d)
    p = fftw.fftwf_alloc_real(h * w * d)
    f = _host_array(p, h * w * d, np.float32)
    f[:] = x.ravel()
    plan = fftw.fftwf_plan_many_r2r(2, _ia([h, w]), d, p, None, d, 1, p, None, d, 1, _ia([5, 5]), 1 << 6)
    assert plan
    fftw.fftwf_execute(plan)
    ref = ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10, impl="port")
    assert np.abs(f.reshape(h, w, d) - ref).max() <= 1e-5 * np.abs(ref).max()
    fftw.fftwf_destroy_plan(plan)
    fftw.fftwf_free(p)


def test_scan_plan_out_of_place_measure_does_not_touch_arrays(fftw):
    # scan/scan.c:352-359: buffers zeroed BEFORE planning with FFTW_MEASURE, input must stay intact after execute
    h, w, c = 64, 96, 3
    n = h * w * c
    pin, pout = fftw.fftwf_alloc_real(n), fftw.fftwf_alloc_real(n)
    a, b = _host_array(pin, n, np.float32), _host_array(pout, n, np.float32)
    a[:] = 0
    b[:] = 7
    plan = fftw.fftwf_plan_many_r2r(2, _ia([h, w]), c, pin, None, c, 1, pout, None, c, 1, _ia([4, 4]), 0)
    assert np.all(a == 0) and np.all(b == 7)          # plan time: untouched
    x = ol.synth_f32(9, n)
    a[:] = x
    fftw.fftwf_execute(plan)
    assert np.array_equal(a, x)
    ref = ol.dct2d_interleaved(x.reshape(h, w, c).astype(np.float64), ol.REDFT01, impl="port")
    assert np.abs(b.reshape(h, w, c) - ref).max() <= 1e-5 * np.abs(ref).max()
    fftw.fftwf_execute(plan)                            # executes repeatedly (scan.c:447)
    assert np.abs(b.reshape(h, w, c) - ref).max() <= 1e-5 * np.abs(ref).max()
    fftw.fftwf_destroy_plan(plan)
    fftw.fftwf_free(pin)
    fftw.fftwf_free(pout)


def test_motion_plan_embedded_3d_host(fftw, golden):
    d, h, w, md, mh, mw = [int(v) for v in golden["vol_dims"]]
    n = md * mh * mw
    p = fftw.fftwf_alloc_real(n)
    f = _host_array(p, n, np.float32)
    f[:] = golden["vol_in"].astype(np.float32).ravel()
    plan = fftw.fftwf_plan_many_r2r(3, _ia([d, h, w]), 1, p, _ia([md, mh, mw]), 1, 0, p, _ia([md, mh, mw]), 1, 0, _ia([5, 5, 5]), 1 << 6)
    fftw.fftwf_execute(plan)
    ref = golden["vol_redft10"]
    assert np.abs(f.reshape(md, mh, mw) - ref).max() <= 1e-5 * np.abs(ref).max()
    fftw.fftwf_destroy_plan(plan)
    fftw.fftwf_free(p)


def test_double_precision_entry_points(fftw):
    h, w, d = 48, 64, 3
    x = ol.synth_f32(5, h * w * d).astype(np.float64)
    p = fftw.fftw_alloc_real(h * w * d)
    f = _host_array(p, h * w * d, np.float64)
    f[:] = x
    plan = fftw.fftw_plan_many_r2r(2, _ia([h, w]), d, p, None, d, 1, p, None, d, 1, _ia([5, 5]), 1 << 6)
    fftw.fftw_execute(plan)
    ref = ol.dct2d_interleaved(x.reshape(h, w, d), ol.REDFT10)
    # the fftw_ entry points compute in double on the device (spec's default build, precision.h:50-53)
    assert np.abs(f.reshape(h, w, d) - ref).max() <= 1e-13 * np.abs(ref).max()
    fftw.fftw_destroy_plan(plan)
    # out of place + inverse: REDFT01 of the coefficients returns 4hw * x and leaves the input alone
    q = fftw.fftw_alloc_real(h * w * d)
    g = _host_array(q, h * w * d, np.float64)
    g[:] = np.nan
    keep = f.copy()
    plan = fftw.fftw_plan_many_r2r(2, _ia([h, w]), d, p, None, d, 1, q, None, d, 1, _ia([4, 4]), 1 << 6)
    fftw.fftw_execute(plan)
    assert np.array_equal(f, keep)
    assert np.abs(g / (4.0 * h * w) - x).max() <= 1e-14
    fftw.fftw_destroy_plan(plan)
    fftw.fftw_free(q)
    fftw.fftw_free(p)


def _bind_r2r_2d(lib):
    for name in ("fftwf_plan_r2r_2d", "fftw_plan_r2r_2d"):
        f = getattr(lib, name)
        f.restype = C.c_void_p
        f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint]


def test_plan_r2r_2d_in_place_float(fftw):
    # applybasis/draw.c:66-76 exactly: alloc_real, zeroed canvas, a few coefficients, plan_r2r_2d(h, w, c, c, REDFT01, REDFT01, ESTIMATE), execute
    _bind_r2r_2d(fftw)
    h, w = 96, 160
    p = fftw.fftwf_alloc_real(h * w)
    f = _host_array(p, h * w, np.float32)
    f[:] = 0
    for (x, y, v) in ((3, 2, 0.5), (10, 7, 0.25), (0, 5, 0.125), (159, 95, -0.3)):
        f[y * w + x] = v / 4
    f[0] += 0.5
    src = f.copy()
    plan = fftw.fftwf_plan_r2r_2d(h, w, p, p, ol.REDFT01, ol.REDFT01, 1 << 6)
    assert plan
    assert np.array_equal(f, src)                       # plan time: untouched
    fftw.fftwf_execute(plan)
    ref = ol.r2r_many(src.astype(np.float64), [h, w], [ol.REDFT01, ol.REDFT01])
    assert np.abs(f - ref).max() <= 1e-5 * np.abs(ref).max()
    fftw.fftwf_destroy_plan(plan)
    fftw.fftwf_free(p)


@pytest.mark.parametrize("kinds", [(ol.REDFT10, ol.REDFT10), (ol.REDFT01, ol.REDFT10), (ol.REDFT10, ol.REDFT01)])
def test_plan_r2r_2d_out_of_place_double_mixed_kinds(fftw, kinds):
    # the basic interface takes one kind per axis (FFTW manual 4.3.5); out of place leaves the input alone
    _bind_r2r_2d(fftw)
    h, w = 48, 80
    x = ol.synth_f32(17, h * w).astype(np.float64) - 0.5
    p, q = fftw.fftw_alloc_real(h * w), fftw.fftw_alloc_real(h * w)
    f, g = _host_array(p, h * w, np.float64), _host_array(q, h * w, np.float64)
    f[:] = x
    g[:] = np.nan
    plan = fftw.fftw_plan_r2r_2d(h, w, p, q, kinds[0], kinds[1], 1 << 6)
    assert plan
    fftw.fftw_execute(plan)
    assert np.array_equal(f, x)
    ref = ol.r2r_many(x, [h, w], list(kinds))
    assert np.abs(g - ref).max() <= 1e-13 * np.abs(ref).max()
    fftw.fftw_destroy_plan(plan)
    fftw.fftw_free(q)
    fftw.fftw_free(p)


def _read_p1f(path, dtype):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P1F"
        w, h = [int(v) for v in f.readline().split()]
        return np.frombuffer(f.read(), dtype=dtype).reshape(h, w)


@pytest.mark.parametrize("exe,dtype,tol", [("draw_gpu", np.float32, 1e-5), ("draw_gpu_d", np.float64, 1e-13)])
def test_draw_harness(tmp_path, exe, dtype, tol):
    """applybasis/draw.c:43-76 through the C harness: -f components (one without a strength: it takes what the others leave of 1)
    on a zeroed canvas, + 0.5 at DC, REDFT01 x REDFT01 through fftw(plan_r2r_2d)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    w, h = 200, 120
    out = str(tmp_path / "canvas.raw")
    comps = ["3x2:0.5", "10x7", "0x5:0.125", "199x119:-0.25"]
    subprocess.check_call([os.path.join(ROOT, "host", exe), "%dx%d" % (w, h), out] + comps)
    img = _read_p1f(out, dtype)
    c = np.zeros((h, w), dtype=dtype)
    energy = dtype(0.5) + dtype(0.125) + dtype(-0.25)
    for spec, v in (((3, 2), 0.5), ((10, 7), None), ((0, 5), 0.125), ((199, 119), -0.25)):
        c[spec[1], spec[0]] = (dtype(v) if v is not None else (dtype(1) - energy) / dtype(1)) / dtype(4)
    c[0, 0] += dtype(0.5)
    ref = ol.r2r_many(c.astype(np.float64), [h, w], [ol.REDFT01, ol.REDFT01]).reshape(h, w)
    assert np.abs(img - ref).max() <= tol * np.abs(ref).max()


def _write_ppm(path, img_u8):
    h, w, _ = img_u8.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(img_u8.tobytes())


def _read_pf(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = [int(v) for v in f.readline().split()]
        f.readline()
        return np.frombuffer(f.read(), dtype=np.float32).reshape(h, w, 3)


def test_c1_spec_ispec_harness_roundtrip(tmp_path):
    """BASELINE config 1: 256x256 3-channel PPM through the C harness (spec.c:63-78, ispec.c:153-167)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    img = ol.synth_u8(0xD5F0001, 256 * 256 * 3).reshape(256, 256, 3)
    ppm, spec, back = tmp_path / "in.ppm", tmp_path / "spec.pf", tmp_path / "back.pf"
    _write_ppm(ppm, img)
    exe = os.path.join(ROOT, "host", "spec_gpu")
    subprocess.check_call([exe, "spec", str(ppm), str(spec)])
    subprocess.check_call([exe, "ispec", str(spec), str(back)])
    x = img.astype(np.float64) / 255.0
    ref = np.ascontiguousarray(ol.dct2d_interleaved(x, ol.REDFT10))
    ol.lib().oracle_spec_normalise_f64(ref.ctypes.data, 256, 256, 3)
    got = _read_pf(spec)
    assert np.abs(got - ref).max() <= 1e-6                       # uniform range [-1,1]
    assert np.abs(_read_pf(back) - x).max() <= 1e-6              # roundtrip <= 1e-6 abs


def _read_pd(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PD"
        w, h = [int(v) for v in f.readline().split()]
        f.readline()
        return np.frombuffer(f.read(), dtype=np.float64).reshape(h, w, 3)


def test_c1_spec_ispec_harness_double_build(tmp_path):
    """the same harness built with COEFF_PRECISION=D -- the reference's default for spec (spec/Makefile:1): coeff is
    double, fftw(call) is fftw_call, and the engine computes in double"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    img = ol.synth_u8(0xD5F0001, 256 * 256 * 3).reshape(256, 256, 3)
    ppm, spec, back = tmp_path / "in.ppm", tmp_path / "spec.pd", tmp_path / "back.pd"
    _write_ppm(ppm, img)
    exe = os.path.join(ROOT, "host", "spec_gpu_d")
    subprocess.check_call([exe, "spec", str(ppm), str(spec)])
    subprocess.check_call([exe, "ispec", str(spec), str(back)])
    x = img.astype(np.float64) / 255.0
    ref = np.ascontiguousarray(ol.dct2d_interleaved(x, ol.REDFT10))
    ol.lib().oracle_spec_normalise_f64(ref.ctypes.data, 256, 256, 3)
    assert np.abs(_read_pd(spec) - ref).max() <= 1e-14
    assert np.abs(_read_pd(back) - x).max() <= 1e-13


def test_scan_harness(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    img = ol.synth_u8(0xD5F0004, 96 * 64 * 3).reshape(64, 96, 3)
    ppm, out = tmp_path / "in.ppm", tmp_path / "sum.pf"
    _write_ppm(ppm, img)
    r = subprocess.run([os.path.join(ROOT, "host", "scan_gpu"), str(ppm), str(out), "500"], stderr=subprocess.PIPE, check=True)
    assert np.abs(_read_pf(out) - img.astype(np.float64) / 255.0).max() <= 5e-6
    errs = [float(ln.split("=")[1]) for ln in r.stderr.decode().splitlines() if "max|sum-input|" in ln]
    assert len(errs) == (96 * 64 + 499) // 500 and errs[-1] <= 5e-6 and errs[0] > errs[-1]


def _y4m(path, frames_yuv, w, h):
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F25:1 Ip A1:1 C420jpeg\n" % (w, h))
        for fr in frames_yuv:
            f.write(b"FRAME\n")
            f.write(fr.tobytes())


def _read_y4m(path, fb):
    raw = open(path, "rb").read()
    body = raw[raw.index(b"\n") + 1:]
    out = []
    while body:
        assert body.startswith(b"FRAME\n")
        out.append(np.frombuffer(body[6:6 + fb], dtype=np.uint8))
        body = body[6 + fb:]
    return out


@pytest.mark.parametrize("depth", [1, 4, 0])
def test_motion_harness_y4m_roundtrip(tmp_path, depth):
    """config 5's pipeline at small size: yuv420p Y4M through motion's block loop (2-D per frame, 3-D blocks of 4,
    whole clip as one 3-D block) reproduces every 8-bit sample (motion.c:617-776 with no filter)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    w, h, n = 96, 54, 8                     # chroma 48x27
    fb = w * h + 2 * ((w + 1) // 2) * ((h + 1) // 2)
    frames = [ol.synth_u8(0xD5F0005 + i, fb) for i in range(n)]
    src, dst = tmp_path / "in.y4m", tmp_path / "out.y4m"
    _y4m(src, frames, w, h)
    subprocess.check_call([os.path.join(ROOT, "host", "motion_gpu"), str(src), str(dst), str(depth)])
    got = _read_y4m(dst, fb)
    assert len(got) == n
    for a, b in zip(got, frames):
        assert np.array_equal(a, b)


def test_motion_harness_quantiser_matches_oracle(tmp_path):
    """-q: uniform-range coefficients rounded to multiples of q*8*sqrt(N) (motion.c:570,740-744), checked on the luma
    plane of a single 3-D block against the f64 restatement"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    w, h, n, q = 64, 32, 4, 0.02
    fb = w * h + 2 * (w // 2) * (h // 2)
    frames = [ol.synth_u8(77 + i, fb) for i in range(n)]
    src, dst = tmp_path / "in.y4m", tmp_path / "out.y4m"
    _y4m(src, frames, w, h)
    subprocess.check_call([os.path.join(ROOT, "host", "motion_gpu"), str(src), str(dst), "0", str(q)])
    got = np.stack([g[:w * h] for g in _read_y4m(dst, fb)]).reshape(n, h, w)
    vol = np.stack([f[:w * h] for f in frames]).astype(np.float64).reshape(n, h, w)
    c = ol.r2r_many(vol, [n, h, w], [ol.REDFT10] * 3, impl="port")
    ol.lib().oracle_motion_uniform_f64(c.ctypes.data, n, h, w, h, w, 1)
    Q = np.float32(q * 8 * np.sqrt(float(w * h * n)))
    c = np.round(c / Q) * Q
    ol.lib().oracle_motion_uniform_f64(c.ctypes.data, n, h, w, h, w, -1)
    back = ol.r2r_many(c, [n, h, w], [ol.REDFT01] * 3, impl="port").reshape(n, h, w) / (8.0 * w * h * n)
    ref = np.clip(np.round(back), 0, 255).astype(np.uint8)
    # coefficients sitting exactly on a rounding boundary may flip between f32 and f64: allow a handful of +-1 LSB pixels
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert diff.max() <= 2 and (diff > 0).mean() < 0.02


@pytest.mark.parametrize("method,step", [("diagonal", 9), ("mirror", 5), ("radial", 7), ("horizontal", 700), ("column", 11)])
def test_scan_harness_other_methods(tmp_path, method, step):
    """every permutation scan method reconstructs the image exactly once all indices are summed (scan.c:377-459)"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    img = ol.synth_u8(31, 80 * 48 * 3).reshape(48, 80, 3)
    ppm, out = tmp_path / "in.ppm", tmp_path / "sum.pf"
    _write_ppm(ppm, img)
    subprocess.run([os.path.join(ROOT, "host", "scan_gpu"), str(ppm), str(out), str(step), method], stderr=subprocess.PIPE, check=True)
    assert np.abs(_read_pf(out) - img.astype(np.float64) / 255.0).max() <= 5e-6


@pytest.mark.parametrize("seed", range(16))
def test_random_geometries_on_host_pointers(fftw, seed):
    """random advanced-interface geometries through fftwf_/fftw_plan_many_r2r on pinned host arrays: the owned elements
    match the definition, everything else in the arrays is left as it was"""
    import test_kernel_logic_cpu as tk
    rng = np.random.default_rng(7000 + seed)
    rank, n, howmany, embed, stride, dist, total, kinds = tk._random_case(rng)
    f64 = bool(seed % 2)
    oop = bool(rng.integers(0, 2))
    dt = np.float64 if f64 else np.float32
    alloc, free_ = (fftw.fftw_alloc_real, fftw.fftw_free) if f64 else (fftw.fftwf_alloc_real, fftw.fftwf_free)
    mk, ex, kill = (fftw.fftw_plan_many_r2r, fftw.fftw_execute, fftw.fftw_destroy_plan) if f64 else (fftw.fftwf_plan_many_r2r, fftw.fftwf_execute, fftw.fftwf_destroy_plan)
    x = ol.synth_f32(seed + 11, total).astype(dt)
    pin = alloc(total)
    a = _host_array(pin, total, dt)
    a[:] = x
    if oop:
        pout = alloc(total)
        b = _host_array(pout, total, dt)
        b[:] = 7
    else:
        pout, b = pin, a
    plan = mk(rank, _ia(n), howmany, pin, _ia(embed), stride, dist, pout, _ia(embed), stride, dist, _ia(kinds), 1 << 6)
    assert plan
    ex(plan)
    ref_full = ol.r2r_many(x.astype(np.float64), n, kinds, howmany=howmany, inembed=embed, istride=stride, idist=dist, onembed=embed, ostride=stride, odist=dist)
    idx = np.zeros(1, dtype=np.int64)
    mult = stride
    for ax in range(rank - 1, -1, -1):
        idx = (idx[None, :] + (np.arange(n[ax]) * mult)[:, None]).ravel()
        mult *= embed[ax]
    idx = (idx[None, :] + (np.arange(howmany) * dist)[:, None]).ravel()
    ref = np.full(total, 7.0) if oop else x.astype(np.float64).copy()
    ref[idx] = ref_full[idx]
    if oop:
        assert np.array_equal(a, x)
    scale = max(np.abs(ref_full[idx]).max(), 1e-30)
    assert np.abs(b.astype(np.float64) - ref).max() <= (5e-13 if f64 else 3e-6) * scale, (n, howmany, embed, stride, dist, kinds, oop)
    kill(plan)
    free_(pin)
    if oop:
        free_(pout)


@pytest.mark.parametrize("method", ["horizontal", "vertical", "zigzag", "row", "column", "diagonal", "mirror", "box", "ibox", "radial", "iradial",
                                    "magnitude", "magnitude:200", "file:coordinate", "file:index", "file:box"])
def test_scan_device_resident_harness_every_method(tmp_path, method):
    """host/scan_dev.c: scan's loop with every buffer and every scan order on the GPU.  The final sum equals
    the input for every method that visits each pixel once; box (shared pixels are added once per frame they appear in) is checked
    against the host-pointer harness scan_gpu, which runs the reference's own loop shape."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    w, h = 80, 48
    img = ol.synth_u8(31, w * h * 3).reshape(h, w, 3)
    ppm, out = tmp_path / "in.ppm", tmp_path / "sum.pf"
    _write_ppm(ppm, img)
    arg, step = method, "7"
    if method.startswith("file:"):
        kind = method.split(":")[1]
        import ctypes as C
        so = C.CDLL(os.path.join(ROOT, "host", "libscanorders.so"))
        libc = C.CDLL(None)
        libc.fopen.restype = C.c_void_p
        libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        libc.fclose.argtypes = [C.c_void_p]
        so.scan_order_serialize_coordinate.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
        so.scan_order_serialize_index.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
        path = tmp_path / "order.txt"
        f = libc.fopen(str(path).encode(), b"w")
        if kind == "box":        # wide frame: in range, but pixels shared between indices -> per-frame lists + stamps
            assert so.scan_order_serialize_coordinate(7, w, h, f) == 0
        elif kind == "coordinate":
            assert so.scan_order_serialize_coordinate(6, w, h, f) == 0        # mirror
        else:
            assert so.scan_order_serialize_index(5, w, h, f) == 0             # diagonal
        libc.fclose(f)
        arg = "file:" + str(path)
    r = subprocess.run([os.path.join(ROOT, "host", "scan_dev"), str(ppm), str(out), step, arg], stderr=subprocess.PIPE, check=True)
    assert b"device-resident" in r.stderr
    got = _read_pf(out)
    if method in ("box", "file:box"):
        ref_out = tmp_path / "ref.pf"
        subprocess.run([os.path.join(ROOT, "host", "scan_gpu"), str(ppm), str(ref_out), step, "box"], stderr=subprocess.PIPE, check=True)
        assert np.abs(got - _read_pf(ref_out)).max() <= 5e-6
    else:
        assert np.abs(got - img.astype(np.float64) / 255.0).max() <= 5e-6


def _zoom_oracle(x, btype, xs, ys, vx, vy, vw, vh):
    """zoom.c:347-375 restated in f64 (oracle/): basis x coefficients x basis of the image's REDFT10^2"""
    import ctypes as C
    h, w, _ = x.shape
    L = ol.lib()
    cf = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10))
    cw = L.oracle_zoom_basis_f64(None, btype, xs[0], xs[1], vx, vw, w)
    ch = L.oracle_zoom_basis_f64(None, btype, ys[0], ys[1], vy, vh, h)
    xb = np.zeros(max(1, vw * (cw - 1))); yb = np.zeros(max(1, vh * (ch - 1)))
    L.oracle_zoom_basis_f64(xb.ctypes.data, btype, xs[0], xs[1], vx, vw, w)
    L.oracle_zoom_basis_f64(yb.ctypes.data, btype, ys[0], ys[1], vy, vh, h)
    ref = np.zeros((vh, vw, 3))
    L.oracle_zoom_product_f64(cf.ctypes.data, w, h, xb.ctypes.data, cw, yb.ctypes.data, ch, ref.ctypes.data, vw, vh)
    return ref


def _read_pfs(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PFS"
        vw, vh, n = [int(v) for v in f.readline().split()]
        f.readline()
        data = np.frombuffer(f.read(), dtype=np.float32)
    return data.reshape(-1, vh, vw, 3), n


@pytest.mark.parametrize("btype,xs,ys,pos,view,method,how", [
    (0, (2.0, 1.0), (2.0, 1.0), (0.0, 0.0), None, "auto", "fft"),                     # integer scale on the DCT-III grid
    (0, (3.0, 2.0), (5.0, 4.0), (3.5, 1.25), (50, 30), "auto", "fft"),                # rational scales with integer scaled lengths (72 x 45), a view, an offset
    (0, (7.0, 5.0), (5.0, 4.0), (3.5, 1.25), (50, 30), "auto", "czt"),                # 48 x 7/5 = 67.2 samples: off the DCT-III grid
    (1, (1.7, 1.0), (1.7, 1.0), (0.0, 0.0), None, "auto", "czt"),
    (2, (2.0, 1.0), (2.0, 1.0), (28.0, 26.0), (40, 20), "auto", None),                # a centred view: ((96 - 40) / 2, (72 - 20) / 2)
    (0, (1.5, 1.0), (1.5, 1.0), (0.0, 0.0), None, "gemm", "gemm"),                    # the dense product, forced
])
def test_zoom_harness(tmp_path, btype, xs, ys, pos, view, method, how):
    """zoom/zoom.c:263-265,347-375 through the C harness (host/zoom_gpu.c): fftw(plan_many_r2r) REDFT10^2 on the host buffer as the tool calls
    it, then one frame by whichever of the three device paths applies -- against the f64 restatement.  The frame's geometry is passed as numbers
    (the tool's option handling is not part of the path)"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    w, h = 48, 36
    x = ol.synth_f32(4321, w * h * 3).reshape(h, w, 3)
    src = tmp_path / "in.pf"
    with open(src, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h)); f.write(x.astype(np.float32).tobytes())
    out = tmp_path / "out.raw"
    vw, vh = view if view else (int(w * xs[0] / xs[1]), int(h * ys[0] / ys[1]))           # zoom.c:286-289
    args = [str(src), str(out), str(btype), repr(xs[0]), repr(xs[1]), repr(ys[0]), repr(ys[1]), repr(pos[0]), repr(pos[1]), str(vw), str(vh), method]
    r = subprocess.run([os.path.join(ROOT, "host", "zoom_gpu")] + args, stderr=subprocess.PIPE, check=True)
    frames, n = _read_pfs(out)
    assert n == 1 and frames.shape == (1, vh, vw, 3)
    if how:
        assert (" by " + how).encode() in r.stderr, r.stderr
    ref = _zoom_oracle(x, btype, xs, ys, pos[0], pos[1], vw, vh)
    assert np.abs(frames[0] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def _band(h, w, c, frame, nframes, rng):
    """what scan/scan.c:429-445 leaves in `reconstruction` for one output frame of a zigzag scan: zeros but for an anti-diagonal band"""
    a = np.zeros((h, w, c), np.float32)
    y, x = np.mgrid[0:h, 0:w]
    d = (y * w // h + x)
    lo, hi = (2 * w) * frame // nframes, (2 * w) * (frame + 1) // nframes
    m = (d >= lo) & (d < hi)
    a[m] = rng.standard_normal((int(m.sum()), c)).astype(np.float32)
    return a


@pytest.mark.parametrize("h,w,c,inplace,dtype", [(1021, 3001, 3, False, np.float32), (1024, 3072, 3, True, np.float32), (511, 3071, 3, False, np.float64)])
def test_sparse_upload_delivers_the_bytes_of_the_dense_copy(fftw, monkeypatch, h, w, c, inplace, dtype):
    """scan/scan.c:429-447: `reconstruction` is mostly zeros at every execute.  Arrays of >= 32 MB go up as packed non-zero 4 KB blocks; the output must be
    bit for bit what the dense copy gives -- frame after frame through ONE plan (nothing of the previous frame may linger), with a -0.0f that is the only
    thing in its block, an array whose byte count is not a multiple of the block, a dense frame in between (dense copy, then the look-ahead backs off)."""
    fftw.dspfft_fftw_sparse_uploads.restype = C.c_ulonglong
    n = h * w * c
    es = np.dtype(dtype).itemsize
    f64 = dtype == np.float64
    alloc, mkplan, execute, destroy, free = ((fftw.fftw_alloc_real, fftw.fftw_plan_many_r2r, fftw.fftw_execute, fftw.fftw_destroy_plan, fftw.fftw_free) if f64 else
                                             (fftw.fftwf_alloc_real, fftw.fftwf_plan_many_r2r, fftw.fftwf_execute, fftw.fftwf_destroy_plan, fftw.fftwf_free))
    bits = np.uint64 if f64 else np.uint32
    assert n * es >= 32 << 20 and (inplace or (n * es) % 4096)
    pin = alloc(n)
    pout = pin if inplace else alloc(n)
    a, b = _host_array(pin, n, dtype), _host_array(pout, n, dtype)
    a[:] = 0
    plan = mkplan(2, _ia([h, w]), c, pin, None, c, 1, pout, None, c, 1, _ia([4, 4]), 1 << 6)
    assert plan
    rng = np.random.default_rng(h)
    frames = [_band(h, w, c, f, 48, rng).ravel().astype(dtype) for f in (0, 23, 47)]       # a band is 1/24 of every row it crosses: <= 2 of its 9 blocks
    frames[1][-1] = 1.5                                             # the tail that does not fill a block
    lone = np.zeros(n, dtype); lone[n // 2] = -0.0; lone[7] = 2.0        # a block whose only set bit is a sign
    dense = rng.standard_normal(n).astype(dtype)
    order = [frames[0], frames[1], lone, dense, frames[2], frames[0]]

    def run_all():
        outs = []
        for x in order:
            a[:] = x
            execute(plan)
            outs.append(b.copy())
        return outs
    monkeypatch.setenv("DSPFFT_UPLOAD_THREADS", "0")
    before = fftw.dspfft_fftw_sparse_uploads()
    want = run_all()
    assert fftw.dspfft_fftw_sparse_uploads() == before              # 0 threads: the dense copy, always
    monkeypatch.setenv("DSPFFT_UPLOAD_THREADS", "6")
    got = run_all()
    # frames 0, 1, lone go up sparse; `dense` fills a thread's share and goes up whole; the next execute is not looked at (back-off 1); the last is
    assert fftw.dspfft_fftw_sparse_uploads() == before + 4
    for g, w_ in zip(got, want):
        assert np.array_equal(g.view(bits), w_.view(bits))
    # and against the oracle, once
    a[:] = frames[1]
    execute(plan)
    ref = ol.dct2d_interleaved(frames[1].reshape(h, w, c).astype(np.float64), ol.REDFT01, impl="port")
    assert np.abs(b.reshape(h, w, c) - ref).max() <= (1e-12 if f64 else 1e-5) * np.abs(ref).max()
    destroy(plan)
    free(pin)
    if not inplace:
        free(pout)
