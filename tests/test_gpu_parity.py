"""GPU (-m gpu): parity of the HIP product library (through the C ABI of include/dspfft.h) against
the oracle.  float tolerance from BASELINE.json north_star: max|gpu-ref| <= 1e-5 * max|ref| (and
rms <= 1e-5 * rms(ref)); integer scan order bit-exact."""
import os
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dspfun_amd import _lib
    _lib.load()   # fails loudly if the HIP library was not built
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def check(got, ref, tol=TOL):
    got = got.astype(np.float64)
    m = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
    r = np.sqrt(np.mean((got - ref) ** 2)) / max(np.sqrt(np.mean(ref ** 2)), 1e-30)
    assert m <= tol and r <= tol, (m, r)


def plan_image(h, w, c, kind):
    from dspfun_amd import Plan
    return Plan.image(h, w, c, kind)


@pytest.mark.parametrize("i", range(8))
def test_golden_images(gpu, golden, i):
    from dspfun_amd import REDFT10, REDFT01
    x = golden[f"img{i}_in"]
    h, w, c = x.shape
    for kind, name in ((REDFT10, "redft10"), (REDFT01, "redft01")):
        d = dev(gpu, x)
        plan_image(h, w, c, kind).execute(d.data_ptr())
        gpu.cuda.synchronize()
        check(d.cpu().numpy(), golden[f"img{i}_{name}"])


@pytest.mark.parametrize("N", (2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 17, 27, 30, 31, 45, 60, 64, 97, 135, 270, 540))
def test_golden_vectors(gpu, golden, N):
    from dspfun_amd import Plan, REDFT10, REDFT01
    x = golden[f"vec{N}_in"]
    for kind, name in ((REDFT10, "redft10"), (REDFT01, "redft01")):
        d = dev(gpu, x)
        Plan.many_r2r([N], [kind]).execute(d.data_ptr())
        gpu.cuda.synchronize()
        check(d.cpu().numpy(), golden[f"vec{N}_{name}"])


def test_golden_volume_embedded(gpu, golden):
    from dspfun_amd import Plan, REDFT10, REDFT01
    d_, h, w, md, mh, mw = [int(v) for v in golden["vol_dims"]]
    buf = golden["vol_in"].astype(np.float32)
    for kind, name in ((REDFT10, "redft10"), (REDFT01, "redft01")):
        d = dev(gpu, buf)
        Plan.many_r2r([d_, h, w], [kind] * 3, inembed=[md, mh, mw], onembed=[md, mh, mw]).execute(d.data_ptr())
        gpu.cuda.synchronize()
        got = d.cpu().numpy()
        check(got, golden[f"vol_{name}"])
        mask = np.ones((md, mh, mw), bool); mask[:d_, :h, :w] = False
        assert np.array_equal(got[mask], buf[mask])


@pytest.mark.parametrize("h,w,c", [(256, 256, 3), (270, 480, 3), (135, 240, 3), (540, 960, 1), (17, 40, 3), (31, 33, 3), (64, 50, 4), (7, 13, 2)])
def test_images_vs_oracle_port(gpu, h, w, c):
    """C1-like sizes, checked against the f64 O(N log N) port (itself pinned to the definition)."""
    from dspfun_amd import REDFT10, REDFT01
    x = ol.synth_f32(0xD5F0001 + h * w, h * w * c).reshape(h, w, c)
    for kind in (REDFT10, REDFT01):
        d = dev(gpu, x)
        plan_image(h, w, c, kind).execute(d.data_ptr())
        gpu.cuda.synchronize()
        check(d.cpu().numpy(), ol.dct2d_interleaved(x.astype(np.float64), kind, impl="port"))


def test_out_of_place_keeps_input(gpu):
    from dspfun_amd import REDFT01
    h, w, c = 96, 160, 3
    x = ol.synth_f32(5, h * w * c).reshape(h, w, c)
    d = dev(gpu, x)
    o = gpu.full_like(d, float("nan"))
    plan_image(h, w, c, REDFT01).execute(d.data_ptr(), o.data_ptr())
    gpu.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), x)
    check(o.cpu().numpy(), ol.dct2d_interleaved(x.astype(np.float64), REDFT01, impl="port"))


def test_c2_full_size_4k(gpu):
    """BASELINE config 2: 3840x2160x3 f32, DCT-II in place then DCT-III in place, x 1/(4wh).
    Forward coefficients vs the f64 port (8 threads, a few seconds); roundtrip <= 5e-6 abs; also the
    zero-mean variant."""
    from dspfun_amd import REDFT10, REDFT01
    h, w, c = 2160, 3840, 3
    base = ol.synth_f32(0xD5F0002, h * w * c).reshape(h, w, c)
    fwd = plan_image(h, w, c, REDFT10)
    inv = plan_image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * w * h))
    for shift in (0.0, 0.5):
        x = (base - np.float32(shift)).astype(np.float32)
        d = dev(gpu, x)
        fwd.execute(d.data_ptr())
        gpu.cuda.synchronize()
        ref = ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=8)
        check(d.cpu().numpy(), ref)
        del ref
        inv.execute(d.data_ptr())
        gpu.cuda.synchronize()
        assert np.abs(d.cpu().numpy() - x).max() <= 5e-6


def test_linearity_full_size(gpu):
    """size-independent property at 4K: T(a x + b y) == a T(x) + b T(y)"""
    from dspfun_amd import REDFT10
    h, w, c = 2160, 3840, 3
    x = dev(gpu, ol.synth_f32(1, h * w * c))
    y = dev(gpu, ol.synth_f32(2, h * w * c))
    z = 0.25 * x - 1.5 * y
    p = plan_image(h, w, c, REDFT10)
    for t in (x, y, z):
        p.execute(t.data_ptr())
    gpu.cuda.synchronize()
    lin = 0.25 * x - 1.5 * y
    assert float((z - lin).abs().max() / lin.abs().max()) < 1e-5


def test_zigzag_bit_exact(gpu, scan_golden):
    """scan/scan_methods.c:69-115 on the device, bit for bit: the survey's hashes, and the ones tests/golden/make_ref_fixtures.py takes from the
    reference's scan_methods.c compiled WITHOUT a stand-in header (ref_scan.npz *_fnv1a: BASELINE's 3840x2160 and 7680x4320 among them)"""
    import os
    from dspfun_amd import _lib
    L = _lib.load()
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_scan.npz"))
    want_by_size = dict(scan_golden["zigzag_fnv"])
    for (w, h) in fx["fnv1a_sizes"]:
        v = "%016x" % int(fx[f"zigzag_{w}x{h}_fnv1a"][0])
        assert want_by_size.setdefault(f"{w}x{h}", v) == v, (w, h)          # where both exist they agree
    for key, want in want_by_size.items():
        w, h = [int(v) for v in key.split("x")]
        lin = gpu.zeros(w * h, dtype=gpu.int32, device="cuda:0")
        assert L.dspfft_scan_zigzag(lin.data_ptr(), w, h, 0, w * h, None) == 0
        gpu.cuda.synchronize()
        a = lin.cpu().numpy().view(np.uint32).astype(np.uint64)
        assert "%016x" % ol.lib().oracle_fnv1a64_u64(a.ctypes.data, a.size) == want, key


def test_horizontal_vertical_zigzag_orders_at_8k(gpu):
    """the one-index-per-pixel scan orders at 7680 x 4320 and 3840 x 2160 by the general device generator (dspfft_scan_coords), against the
    hashes of the reference's scan_methods.c compiled without a stand-in header (ref_scan.npz *_fnv1a)"""
    import os
    from dspfun_amd import _lib
    L = _lib.load()
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_scan.npz"))
    for m, name in enumerate(("horizontal", "vertical", "zigzag")):
        for (w, h) in ((7680, 4320), (3840, 2160)):
            lin = gpu.zeros(w * h, dtype=gpu.int32, device="cuda:0")
            assert L.dspfft_scan_coords(lin.data_ptr(), m, w, h, 0, w * h, None) == 0
            gpu.cuda.synchronize()
            a = lin.cpu().numpy().view(np.uint32).astype(np.uint64)
            assert "%016x" % ol.lib().oracle_fnv1a64_u64(a.ctypes.data, a.size) == "%016x" % int(fx[f"{name}_{w}x{h}_fnv1a"][0]), (name, w, h)


def test_scan_frames_c4_like(gpu):
    """scan/scan.c:377-383,421-459 on device at 480x270x3, zigzag, 8 frames: per-frame sum vs the
    f64 restatement and final sum == input."""
    from dspfun_amd import _lib, REDFT10, REDFT01
    L = _lib.load()
    w, h, c = 480, 270, 3
    x = ol.synth_f32(0xD5F0004, w * h * c).reshape(h, w, c)
    coeffs = dev(gpu, x)
    plan_image(h, w, c, REDFT10).set_scale(1.0 / (4.0 * w * h)).execute(coeffs.data_ptr())
    inv = plan_image(h, w, c, REDFT01)
    order = gpu.zeros(w * h, dtype=gpu.int32, device="cuda:0")
    assert L.dspfft_scan_zigzag(order.data_ptr(), w, h, 0, w * h, None) == 0
    total = gpu.empty_like(coeffs)
    recon = gpu.empty_like(coeffs)
    image = gpu.empty_like(coeffs)
    assert L.dspfft_broadcast_dc(total.data_ptr(), coeffs.data_ptr(), w * h, c, None) == 0
    # f64 restatement
    cf64 = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port"))
    ol.lib().oracle_scan_normalise_f64(cf64.ctypes.data, w, h, c)
    ref_total = np.ascontiguousarray(np.broadcast_to(cf64[0, 0], (h, w, c)).copy())
    zz = ol.zigzag_order(w, h)
    nframes = 8
    step = (w * h + nframes - 1) // nframes
    for f in range(nframes):
        first, count = f * step, min(step, w * h - f * step)
        assert L.dspfft_scan_scatter(recon.data_ptr(), coeffs.data_ptr(), order.data_ptr() + 4 * first, count, w * h, c, None) == 0
        inv.execute(recon.data_ptr(), image.data_ptr())
        assert L.dspfft_accumulate(total.data_ptr(), image.data_ptr(), w * h * c, None) == 0
        lin = np.ascontiguousarray(zz[first:first + count])
        # oracle frame (port-based inverse for speed)
        rec = np.zeros_like(cf64)
        ys, xs = (lin // w).astype(np.int64), (lin % w).astype(np.int64)
        rec[ys, xs] = cf64[ys, xs]
        rec[0, 0] = 0
        ref_total += ol.dct2d_interleaved(rec, REDFT01, impl="port")
        gpu.cuda.synchronize()
        assert np.abs(total.cpu().numpy() - ref_total).max() < 5e-6, f
    assert np.abs(total.cpu().numpy() - x).max() <= 5e-6


def test_motion_u8_roundtrip(gpu):
    """motion/motion.c:617-647,748-776 with block == scaled on a 3-D block: u8 in == u8 out."""
    from dspfun_amd import Plan, _lib, REDFT10, REDFT01
    L = _lib.load()
    d_, h, w = 16, 90, 160
    pix = ol.synth_u8(0xD5F0005, d_ * h * w)
    src = dev(gpu, pix)
    c = gpu.empty(d_ * h * w, dtype=gpu.float32, device="cuda:0")
    assert L.dspfft_u8_to_f32(c.data_ptr(), src.data_ptr(), pix.size, None) == 0
    r2 = float(np.sqrt(2.0))
    fwd = Plan.many_r2r([d_, h, w], [REDFT10] * 3).set_scale(2 * r2)
    inv = Plan.many_r2r([d_, h, w], [REDFT01] * 3).set_scale(1.0 / (2 * r2))
    for a in range(3):
        fwd.set_axis_scale0(a, 1.0, 1.0 / r2)
        inv.set_axis_scale0(a, r2, 1.0)
    fwd.execute(c.data_ptr())
    gpu.cuda.synchronize()
    mean = pix.astype(np.float64).mean()
    assert abs(float(c[0]) / (8.0 * d_ * h * w) - mean) < 1e-3      # u[0] * normalization^2 == mean
    inv.execute(c.data_ptr())
    out = gpu.empty(d_ * h * w, dtype=gpu.uint8, device="cuda:0")
    assert L.dspfft_f32_to_u8(out.data_ptr(), c.data_ptr(), 1.0 / (8.0 * d_ * h * w), pix.size, None) == 0
    gpu.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), pix)


def test_c3_size_1080p_forward(gpu):
    """BASELINE config 3's transform: 1920x1080x3 DCT-II (zoom/zoom.c:263), vs the f64 port."""
    from dspfun_amd import REDFT10, REDFT01
    h, w, c = 1080, 1920, 3
    x = ol.synth_f32(0xD5F0003, h * w * c).reshape(h, w, c)
    for kind in (REDFT10, REDFT01):
        d = dev(gpu, x)
        p = plan_image(h, w, c, kind)
        p.execute(d.data_ptr())
        gpu.cuda.synchronize()
        check(d.cpu().numpy(), ol.dct2d_interleaved(x.astype(np.float64), kind, impl="port", threads=8))


def test_c4_size_8k_roundtrip(gpu):
    """BASELINE config 4's frame size, 7680x4320x3 (398 MB): roundtrip identity and DC == 4wh*mean
    (size-independent properties; the O(N log N) f64 port covers 4K and below)."""
    from dspfun_amd import REDFT10, REDFT01
    h, w, c = 4320, 7680, 3
    x = ol.synth_f32(0xD5F0004, h * w * c).reshape(h, w, c)
    d = dev(gpu, x)
    plan_image(h, w, c, REDFT10).execute(d.data_ptr())
    gpu.cuda.synchronize()
    dc = d[0, 0].cpu().numpy().astype(np.float64)
    want = 4.0 * x.astype(np.float64).sum(axis=(0, 1))
    assert np.abs(dc - want).max() / np.abs(want).max() < 1e-5
    plan_image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * w * h)).execute(d.data_ptr())
    gpu.cuda.synchronize()
    assert float((d.cpu() - gpu.from_numpy(x)).abs().max()) <= 5e-6


def test_generic_and_specialised_kernels_agree(gpu):
    """the same 4K frame through the specialised kernels and (misaligned by one float, so the
    specialised path is refused) through the generic kernels"""
    from dspfun_amd import REDFT10
    h, w, c = 2160, 3840, 3
    x = ol.synth_f32(42, h * w * c)
    a = dev(gpu, x)
    big = gpu.zeros(h * w * c + 4, dtype=gpu.float32, device="cuda:0")
    b = big[1:1 + h * w * c]
    b.copy_(a)
    p = plan_image(h, w, c, REDFT10)
    assert "ROW*" in p.describe() and "COL*" in p.describe()
    p.execute(a.data_ptr())
    p.execute(b.data_ptr())
    gpu.cuda.synchronize()
    assert float((a - b).abs().max() / a.abs().max()) < 2e-6


@pytest.mark.parametrize("h,w,c", [(720, 1280, 3), (1440, 2560, 3), (480, 640, 3), (480, 720, 3), (512, 512, 3), (1024, 1024, 3), (2048, 2048, 3),
                                    (2160, 4096, 3), (4096, 4096, 3), (720, 1280, 1), (4096, 4100, 1), (600, 800, 3), (768, 1024, 3), (900, 1440, 3), (1200, 1600, 3),
                                    (1600, 2560, 3), (1800, 3200, 3), (2880, 5120, 3), (960, 1280, 3), (1152, 2048, 1), (2160, 3840, 1), (2160, 4096, 1), (768, 1024, 1)])
def test_common_resolutions_specialised_vs_oracle(gpu, h, w, c):
    """spec_list.h's entries for the common frame sizes: both transforms against the f64 port, and a roundtrip"""
    from dspfun_amd import REDFT10, REDFT01
    x = ol.synth_f32(h + w, h * w * c).reshape(h, w, c)
    fwd, inv = plan_image(h, w, c, REDFT10), plan_image(h, w, c, REDFT01)
    assert fwd.describe().count("*") >= (2 if w != 4100 else 1), fwd.describe()
    assert ("BLUE N=4100" in fwd.describe()) == (w == 4100)      # 4100 = 2^2 5^2 41: Bluestein along x
    d = dev(gpu, x)
    fwd.execute(d.data_ptr())
    gpu.cuda.synchronize()
    check(d.cpu().numpy(), ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=8))
    inv.set_scale(1.0 / (4.0 * h * w)).execute(d.data_ptr())
    gpu.cuda.synchronize()
    assert np.abs(d.cpu().numpy() - x).max() <= 2e-6


def test_fused_scan_step_c4_like(gpu):
    """the fused per-frame step (dspfft_execute_masked_accumulate) against the unfused device path
    and the f64 restatement, 960x540x3, zigzag, 6 frames"""
    from dspfun_amd import _lib, REDFT10, REDFT01
    L = _lib.load()
    w, h, c = 960, 540, 3
    x = ol.synth_f32(0xD5F0004, w * h * c).reshape(h, w, c)
    coeffs = dev(gpu, x)
    plan_image(h, w, c, REDFT10).set_scale(1.0 / (4.0 * w * h)).execute(coeffs.data_ptr())
    inv = plan_image(h, w, c, REDFT01)
    assert "ROW*" in inv.describe() and "COL*" in inv.describe()
    nframes = 6
    step = (w * h + nframes - 1) // nframes
    ids = gpu.zeros(w * h, dtype=gpu.int32, device="cuda:0")
    assert L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, step, None) == 0
    acc = gpu.empty_like(coeffs)
    work = gpu.empty_like(coeffs)
    assert L.dspfft_broadcast_dc(acc.data_ptr(), coeffs.data_ptr(), w * h, c, None) == 0
    cf64 = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port"))
    ol.lib().oracle_scan_normalise_f64(cf64.ctypes.data, w, h, c)
    ref = np.ascontiguousarray(np.broadcast_to(cf64[0, 0], (h, w, c)).copy())
    idh = (ol.zigzag_order(w, h), None)
    frame_of = np.empty(w * h, dtype=np.int64)
    frame_of[idh[0].astype(np.int64)] = np.arange(w * h) // step
    frame_of[0] = -1
    for f in range(nframes):
        inv.execute_masked_accumulate(coeffs.data_ptr(), work.data_ptr(), acc.data_ptr(), ids.data_ptr(), f, c)
        rec = np.where((frame_of.reshape(h, w) == f)[:, :, None], cf64, 0.0)
        ref += ol.dct2d_interleaved(rec, REDFT01, impl="port", threads=8)
        gpu.cuda.synchronize()
        assert np.abs(acc.cpu().numpy() - ref).max() < 5e-6, f
    assert np.abs(acc.cpu().numpy() - x).max() <= 5e-6


# ---- double precision (the fftw_ API: spec / zoom / applybasis default build, include/precision.h:50-53) ----
TOL64 = 1e-13


@pytest.mark.parametrize("h,w,c", [(48, 64, 3), (270, 480, 3), (17, 40, 3), (45, 50, 2), (9, 10, 4), (60, 90, 1)])
@pytest.mark.parametrize("kind", [5, 4])
def test_f64_plan_vs_oracle(gpu, h, w, c, kind):
    from dspfun_amd import Plan
    x = ol.synth_f32(h * 7 + w, h * w * c).astype(np.float64).reshape(h, w, c) + 1e-9
    p = Plan.image(h, w, c, kind, dtype="f64")
    assert "f64" in p.describe()
    d = gpu.from_numpy(x.copy()).to("cuda:0")
    p.execute(d.data_ptr())
    gpu.cuda.synchronize()
    ref = ol.dct2d_interleaved(x, kind, impl="port", threads=8) if h * w > 20000 else ol.dct2d_interleaved(x, kind)
    check(d.cpu().numpy(), ref, tol=TOL64)


@pytest.mark.parametrize("h,w,c", [(512, 512, 3), (540, 960, 3), (1080, 1920, 3), (1080, 1920, 1), (2160, 3840, 1), (720, 1280, 3), (1440, 2560, 3), (2048, 2048, 3),
                                   (1024, 1024, 3), (4096, 4096, 3), (720, 1280, 1)])
@pytest.mark.parametrize("kind", [5, 4])
def test_f64_specialised_kernels_vs_port(gpu, h, w, c, kind):
    """spec_list.h DSPFFT_*_SPECS_F64 (RowSpecT<double> / ColSpecT<double>): against the f64 port, in place and out of place,
    and the same to rounding as the generic double kernels"""
    from dspfun_amd import Plan
    x = ol.synth_f32(h * 3 + w, h * w * c).astype(np.float64).reshape(h, w, c) * (1 + 2.0 ** -31)
    p = Plan.image(h, w, c, kind, dtype="f64")
    assert "ROW* f64" in p.describe() and "COL* f64" in p.describe(), p.describe()
    ref = ol.dct2d_interleaved(x, kind, impl="port", threads=8)
    d = gpu.from_numpy(x.copy()).to("cuda:0")
    o = gpu.empty_like(d)
    p.execute(d.data_ptr(), o.data_ptr())
    p.execute(d.data_ptr())
    gpu.cuda.synchronize()
    check(d.cpu().numpy(), ref, tol=TOL64)
    check(o.cpu().numpy(), ref, tol=TOL64)


def test_f64_8k_frame_listed_kernels(gpu):
    """7680 x 4320 x 3 doubles (an 8K frame of spec / zoom's default double build, 796 MB): rows as channel lines (the only row pass such a
    line has), columns on the listed 4320 x 4 tile.  Forward coefficients against the definition at sampled positions (a matrix-vector
    product per coefficient column, float64 on the host), roundtrip identity, and one timing line (informative)."""
    from dspfun_amd import Plan
    h, w, c = 4320, 7680, 3
    x = ol.synth_f32(0xD5F0004, h * w * c).astype(np.float64).reshape(h, w, c) - 0.5
    fwd = Plan.image(h, w, c, 5, dtype="f64")
    inv = Plan.image(h, w, c, 4, dtype="f64").set_scale(1.0 / (4.0 * w * h))
    assert "ROW* f64 N=7680 C=3" in fwd.describe() and "channel lines" in fwd.describe() and "COL* f64 N=4320" in fwd.describe(), fwd.describe()
    d = gpu.from_numpy(x).to("cuda:0")
    fwd.execute(d.data_ptr())
    gpu.cuda.synchronize()
    yy = np.arange(h) + 0.5
    xx = np.arange(w) + 0.5
    for ky, kx in ((0, 0), (1, 0), (0, 1), (7, 11), (4319, 7679), (2160, 3840), (4000, 13), (5, 7000)):
        col = np.cos(np.pi * xx * kx / w)
        row = np.cos(np.pi * yy * ky / h)
        ref = 4.0 * np.einsum("y,yc->c", row, np.einsum("yxc,x->yc", x, col))
        got = d[ky, kx].cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-11 * (4.0 * h * w) ** 0.5, (ky, kx, got, ref)
    inv.execute(d.data_ptr())
    gpu.cuda.synchronize()
    assert float((d.cpu() - gpu.from_numpy(x)).abs().max()) <= 1e-13
    a, b = gpu.cuda.Event(enable_timing=True), gpu.cuda.Event(enable_timing=True)
    for _ in range(5):
        fwd.execute(d.data_ptr()); inv.execute(d.data_ptr())
    a.record()
    for _ in range(20):
        fwd.execute(d.data_ptr()); inv.execute(d.data_ptr())
    b.record(); gpu.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"8K f64 roundtrip: {ms:.3f} ms = {96.0 * w * h / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s at 96 B/pixel")


@pytest.mark.parametrize("h", [2160, 27, 5])
def test_f64_channel_lines_equal_the_interleaved_kernel(gpu, h, monkeypatch):
    """3840-pixel RGB double lines run one workgroup per (line, channel) (dct_spec.h RowChanSpecT; the three channel lines of a line
    store a third of every cache line each): bit-identical to the interleaved kernel (DSPFFT_ROW_CHAN=0: same butterflies in the same
    order), in place, out of place and through the fused scan step; line counts with and without a tail of chan_work's groups of 8"""
    from dspfun_amd import Plan, _lib
    w, c = 3840, 3
    x = ol.synth_f32(h + 11, h * w * c).astype(np.float64).reshape(h, w, c) * (1 + 2.0 ** -31)
    for kind in (5, 4):
        monkeypatch.setenv("DSPFFT_ROW_CHAN", "1")
        p = Plan.image(h, w, c, kind, dtype="f64").set_scale(0.37).set_axis_scale0(1, 0.5, 0.7)
        assert "as 3 channel lines" in p.describe(), p.describe()
        res = {}
        for on in ("1", "0"):
            monkeypatch.setenv("DSPFFT_ROW_CHAN", on)
            d = gpu.from_numpy(x.copy()).to("cuda:0")
            o = gpu.zeros_like(d)
            p.execute(d.data_ptr(), o.data_ptr())
            p.execute(d.data_ptr())
            gpu.cuda.synchronize()
            res[on] = (d.cpu().numpy(), o.cpu().numpy())
        assert np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1]) and np.array_equal(res["1"][0], res["1"][1])
        assert np.abs(res["1"][0]).max() > 0
    L = _lib.load()
    coeffs = gpu.from_numpy(x.copy()).to("cuda:0")
    Plan.image(h, w, c, 5, dtype="f64").set_scale(1.0 / (4 * w * h)).execute(coeffs.data_ptr())
    ids = gpu.zeros(h * w, dtype=gpu.int32, device="cuda:0")
    assert L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, (h * w + 2) // 3, None) == 0
    inv = Plan.image(h, w, c, 4, dtype="f64")
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("DSPFFT_ROW_CHAN", on)
        acc = coeffs[0, 0].expand(h, w, c).contiguous()
        work = gpu.zeros_like(acc)
        for f in range(3):
            inv.execute_masked_accumulate(coeffs.data_ptr(), work.data_ptr(), acc.data_ptr(), ids.data_ptr(), f, c)
        gpu.cuda.synchronize()
        res[on] = acc.cpu().numpy()
    assert np.array_equal(res["1"], res["0"])
    assert np.abs(res["1"] - x).max() < 1e-13


def test_f64_c2_frame_roundtrip_with_spec_normalisation(gpu):
    """3840x2160x3 in double: spec.c:63-78 then ispec.c:153-167, normalisation fused, against the f64 port and
    by the round trip"""
    from dspfun_amd import Plan
    h, w, c = 2160, 3840, 3
    x = ol.synth_f32(0xD5F0002, h * w * c).astype(np.float64).reshape(h, w, c)
    r2 = np.sqrt(2.0)
    fwd = Plan.image(h, w, c, 5, dtype="f64").set_scale(1.0 / (2.0 * w * h))
    inv = Plan.image(h, w, c, 4, dtype="f64").set_scale(0.5)
    for a in range(2):
        fwd.set_axis_scale0(a, 1.0, 1.0 / r2)
        inv.set_axis_scale0(a, r2, 1.0)
    d = gpu.from_numpy(x).to("cuda:0")
    fwd.execute(d.data_ptr())
    gpu.cuda.synchronize()
    ref = np.ascontiguousarray(ol.dct2d_interleaved(x, 5, impl="port", threads=8))
    ol.lib().oracle_spec_normalise_f64(ref.ctypes.data, w, h, c)
    check(d.cpu().numpy(), ref, tol=TOL64)
    inv.execute(d.data_ptr())
    gpu.cuda.synchronize()
    assert np.abs(d.cpu().numpy() - x).max() <= 1e-13


# ---- lengths with prime factors > 13: Bluestein inside the column pass; the O(N^2) kernel only as the fallback ----
@pytest.mark.parametrize("h,w,c", [(768, 1366, 3), (683, 1031, 1), (37, 40, 3), (1087, 1933, 3), (41, 2731, 1)])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_bluestein_sizes_vs_oracle(gpu, h, w, c, dtype):
    from dspfun_amd import Plan
    f64 = dtype == "f64"
    x = ol.synth_f32(h * 3 + w, h * w * c).reshape(h, w, c)
    x = x.astype(np.float64) * (1 + 2.0 ** -31) if f64 else x
    for kind in (5, 4):
        p = Plan.image(h, w, c, kind, dtype=dtype)
        assert "BLUE" in p.describe() and "DENSE" not in p.describe(), p.describe()
        d = gpu.from_numpy(x.copy()).to("cuda:0")
        p.execute(d.data_ptr())
        gpu.cuda.synchronize()
        ref = ol.dct2d_interleaved(x.astype(np.float64), kind, impl="port", threads=8)
        check(d.cpu().numpy(), ref, tol=2e-13 if f64 else TOL)


def test_dense_fallback_matches_bluestein(gpu):
    import os
    from dspfun_amd import Plan
    h, w, c = 97, 194, 3
    x = ol.synth_f32(11, h * w * c).reshape(h, w, c)
    os.environ["DSPFFT_NO_BLUESTEIN"] = "1"
    try:
        pd = Plan.image(h, w, c, 5)
    finally:
        del os.environ["DSPFFT_NO_BLUESTEIN"]
    pb = Plan.image(h, w, c, 5)
    assert pd.describe().count("DENSE") == 2 and pb.describe().count("BLUE") == 2
    a, b = dev(gpu, x), dev(gpu, x)
    pd.execute(a.data_ptr()); pb.execute(b.data_ptr())
    gpu.cuda.synchronize()
    ref = ol.dct2d_interleaved(x.astype(np.float64), 5, impl="port")
    check(a.cpu().numpy(), ref)
    check(b.cpu().numpy(), ref)


# ---- zoom: dense basis product on the f32 matrix cores ----
@pytest.mark.parametrize("w,h,scale,off,btype", [(12, 10, 3, 0.0, 0), (40, 24, 2, 0.5, 0), (33, 17, 2, 0.0, 2), (24, 40, 1, 0.0, 1), (64, 48, 0.5, 0.0, 0)])
def test_zoom_small_vs_oracle(gpu, w, h, scale, off, btype):
    from dspfun_amd.zoom import Zoom
    x = ol.synth_f32(w * h, w * h * 3).reshape(h, w, 3)
    vw, vh = max(1, int(round(w * scale))), max(1, int(round(h * scale)))
    z = Zoom(gpu, dev(gpu, x))
    got = z.frame(vw, vh, (scale, 1.0), (scale, 1.0), off, off, btype).cpu().numpy()
    L = ol.lib()
    cf = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10))
    cw = L.oracle_zoom_basis_f64(None, btype, scale, 1.0, off, vw, w)
    ch = L.oracle_zoom_basis_f64(None, btype, scale, 1.0, off, vh, h)
    xb = np.zeros(max(1, vw * (cw - 1))); yb = np.zeros(max(1, vh * (ch - 1)))
    L.oracle_zoom_basis_f64(xb.ctypes.data, btype, scale, 1.0, off, vw, w)
    L.oracle_zoom_basis_f64(yb.ctypes.data, btype, scale, 1.0, off, vh, h)
    ref = np.zeros((vh, vw, 3))
    L.oracle_zoom_product_f64(cf.ctypes.data, w, h, xb.ctypes.data, cw, yb.ctypes.data, ch, ref.ctypes.data, vw, vh)
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_c3_zoom_4x_1080p(gpu):
    """BASELINE config 3: 1920x1080 RGB -> 7680x4320, scale 4/1, offset 0, interpolated basis.
    Property: out[::4, ::4] == input (samples coincide at integer scale)."""
    from dspfun_amd.zoom import Zoom
    w, h = 1920, 1080
    x = ol.synth_f32(0xD5F0003, w * h * 3).reshape(h, w, 3)
    z = Zoom(gpu, dev(gpu, x))
    out = z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0))
    gpu.cuda.synchronize()
    sub = out[::4, ::4].cpu().numpy()
    e_sub = np.abs(sub - x).max()
    print(f"\nc3 dense product: max|out[::4, ::4] - in| = {e_sub:.3e} (max|in| = {np.abs(x).max():.4f})")
    assert e_sub <= 1e-5 * np.abs(x).max()               # north_star: 1e-5 relative (whole frame: tests/test_zoom_c3_tolerance_gpu.py)
    # one full output row against the f64 restatement (row 1234 of 4320)
    L = ol.lib()
    cf = ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10, impl="port", threads=8)
    cw = L.oracle_zoom_basis_f64(None, 0, 4.0, 1.0, 0.0, 4 * w, w)
    xb = np.zeros(4 * w * (cw - 1)); L.oracle_zoom_basis_f64(xb.ctypes.data, 0, 4.0, 1.0, 0.0, 4 * w, w)
    j = 1234
    kv = np.arange(1, h)
    ybj = np.cos(np.pi * ((j / 4.0) + 0.5) * kv / h)
    trow = cf[0] / 2 + np.tensordot(ybj, cf[1:], axes=(0, 0))              # (w, 3): sum over v
    XB = np.concatenate([np.full((4 * w, 1), 0.5), xb.reshape(4 * w, cw - 1)], axis=1)
    ref_row = (XB @ trow) / (w * h)
    e_row = np.abs(out[j].cpu().numpy() - ref_row).max()
    print(f"c3 dense product: row {j} max|gpu - ref| = {e_row:.3e} (max|ref| = {np.abs(ref_row).max():.4f})")
    assert e_row <= 1e-5 * np.abs(ref_row).max()


def test_c5_motion_plane_full_size(gpu):
    """BASELINE config 5, luma plane as ONE 3-D block (`-b 0x0x0`): 1920x1080x256 u8 -> REDFT10^3 with
    motion's uniform scaling (motion.c:644-647) -> inverse (:748-753) -> u8 (:756-776): identical."""
    from dspfun_amd import Plan, _lib, REDFT10, REDFT01
    L = _lib.load()
    d_, h, w = 256, 1080, 1920
    g = gpu.Generator(device="cuda:0"); g.manual_seed(0xD5F0005)
    pix = gpu.randint(0, 256, (d_, h, w), dtype=gpu.uint8, device="cuda:0", generator=g)
    c = gpu.empty((d_, h, w), dtype=gpu.float32, device="cuda:0")
    assert L.dspfft_u8_to_f32(c.data_ptr(), pix.data_ptr(), pix.numel(), None) == 0
    r2 = float(np.sqrt(2.0))
    fwd = Plan.many_r2r([d_, h, w], [REDFT10] * 3).set_scale(2 * r2)
    inv = Plan.many_r2r([d_, h, w], [REDFT01] * 3).set_scale(1.0 / (2 * r2))
    for a in range(3):
        fwd.set_axis_scale0(a, 1.0, 1.0 / r2)
        inv.set_axis_scale0(a, r2, 1.0)
    assert fwd.describe().count("*") == 3, fwd.describe()      # all three passes on specialised kernels
    fwd.execute(c.data_ptr())
    gpu.cuda.synchronize()
    mean = float(pix.double().mean())
    assert abs(float(c[0, 0, 0]) / (8.0 * d_ * h * w) - mean) < 1e-3
    inv.execute(c.data_ptr())
    out = gpu.empty_like(pix)
    assert L.dspfft_f32_to_u8(out.data_ptr(), c.data_ptr(), 1.0 / (8.0 * d_ * h * w), pix.numel(), None) == 0
    gpu.cuda.synchronize()
    assert int((out != pix).sum()) == 0


def test_slab_dct3d_single_rank_matches_rank3_plan(gpu):
    from dspfun_amd import Plan, REDFT10
    from dspfun_amd.dist import SlabDCT3D
    d_, h, w = 16, 270, 480
    x = dev(gpu, ol.synth_u8(3, d_ * h * w).astype(np.float32).reshape(d_, h, w))
    r2 = float(np.sqrt(2.0))
    p = Plan.many_r2r([d_, h, w], [REDFT10] * 3).set_scale(2 * r2)
    for a in range(3):
        p.set_axis_scale0(a, 1.0, 1.0 / r2)
    a_ = x.clone(); p.execute(a_.data_ptr())
    eng = SlabDCT3D(d_, h, w)
    b_ = eng.forward(x.clone())
    gpu.cuda.synchronize()
    assert float((a_ - b_).abs().max() / a_.abs().max()) < 2e-6
    back = eng.inverse(b_)
    gpu.cuda.synchronize()
    assert float((back - x).abs().max()) < 1e-3
    # the same through five row pieces (the pipelined exchange's layout; one rank, so the exchange itself is a no-op)
    eng5 = SlabDCT3D(d_, h, w, chunks=5)
    assert eng5.P == 5
    c5 = eng5.forward(x.clone())
    gpu.cuda.synchronize()
    assert float((a_ - c5).abs().max() / a_.abs().max()) < 2e-6
    assert float((eng5.inverse(c5) - x).abs().max()) < 1e-3


@pytest.mark.parametrize("layout", ["auto", "planar"])
def test_channel_sharded_scan_single_rank(gpu, layout):
    """config 4's channel-sharded layout (dspfun_amd.dist.ChannelShardedScan) reconstructs the running sums of scan.c:421-459; world size 1
    here (a rank that owns every plane keeps the image interleaved; "planar" forces the form the other world sizes use), world 2 and 4 in
    tests/test_dist_cpu.py"""
    from dspfun_amd.dist import ChannelShardedScan
    w, h, c = 960, 540, 3
    x = ol.synth_f32(0xD5F0004, w * h * c).reshape(h, w, c)
    step = (w * h + 4) // 5
    eng = ChannelShardedScan(dev(gpu, x), step, layout=layout)
    assert eng.nframes == 5 and eng.mine == [0, 1, 2] and eng.interleaved == (layout == "auto")
    cf64 = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), 5, impl="port"))
    ol.lib().oracle_scan_normalise_f64(cf64.ctypes.data, w, h, c)
    ref = np.ascontiguousarray(np.broadcast_to(cf64[0, 0], (h, w, c)).copy())
    zz = ol.zigzag_order(w, h)
    k = 0
    while eng.next_frame():
        lin = np.ascontiguousarray(zz[k * step:(k + 1) * step])
        if k < 2:      # the direct-sum restatement is O(points * w * h): check the first frames, then the total
            frame_of = np.full(w * h, -2, dtype=np.int64)
            frame_of[lin.astype(np.int64)] = k
            frame_of[0] = -1
            rec = np.where((frame_of.reshape(h, w) == k)[:, :, None], cf64, 0.0)
            ref += ol.dct2d_interleaved(rec, 4, impl="port", threads=8)
            assert np.abs(eng.gather().cpu().numpy() - ref).max() < 5e-6, k
        k += 1
    assert k == 5
    assert np.abs(eng.gather().cpu().numpy() - x).max() <= 5e-6


# ---- applybasis: basis x pixel partial sums on the f32 matrix cores ----
@pytest.mark.parametrize("func", ["dft", "idft", "dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4", "wht", "dht"])
@pytest.mark.parametrize("w,h,terms,psum,off,ortho", [(16, 8, None, (16, 8), (0, 0), True), (16, 8, (5, 3), (4, 2), (1, 2), False), (32, 32, (8, 8), (1, 1), (0, 0), False)])
def test_applybasis_partsums_vs_oracle(gpu, func, w, h, terms, psum, off, ortho):
    from dspfun_amd.applybasis import partsums, FUNCTIONS
    import ctypes as C
    x = (ol.synth_f32(w * 31 + h, w * h * 3).reshape(h, w, 3) * 2 - 1).astype(np.float32)      # shift2 range (applybasis.c:358-360)
    got = partsums(gpu, dev(gpu, x), func, ortho, terms, psum, off).cpu().numpy()
    kw, kh = terms if terms else (w, h)
    ref = np.zeros(got.shape + (2,), dtype=np.float64)
    L = ol.lib()
    L.oracle_applybasis_partsums_f64.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_longlong] * 2
    x64 = np.ascontiguousarray(x.astype(np.float64))
    L.oracle_applybasis_partsums_f64(ref.ctypes.data, x64.ctypes.data, w, h, FUNCTIONS.index(func), int(ortho), kw, kh, psum[0], psum[1], off[0], off[1])
    refc = ref[..., 0] + 1j * ref[..., 1]
    assert np.abs(got - refc).max() <= 1e-5 * max(1.0, np.abs(refc).max())


# ---- elementwise stages either side of the transform ----
@pytest.mark.parametrize("rangetype", [0, 1, 2])
@pytest.mark.parametrize("scaletype", [0, 1])
@pytest.mark.parametrize("signtype", [0, 1, 2, 3])
def test_spec_encode_decode_vs_oracle(gpu, rangetype, scaletype, signtype):
    import ctypes as C
    from dspfun_amd import _lib, Plan, REDFT10
    L = _lib.load()
    h, w, d = 45, 64, 3
    x = ol.synth_f32(rangetype * 10 + scaletype * 5 + signtype, h * w * d).reshape(h, w, d)
    r2 = float(np.sqrt(2.0))
    f = dev(gpu, x)
    Plan.image(h, w, d, REDFT10).set_scale(1.0 / (2 * w * h)).set_axis_scale0(0, 1, 1 / r2).set_axis_scale0(1, 1, 1 / r2).execute(f.data_ptr())
    gpu.cuda.synchronize()
    coeffs = f.cpu().numpy().copy()                       # uniform-range coefficients (spec.c:78)
    gain = 127.5 * np.sqrt(4.0 * w * h)                   # gaintype native (spec.c:84)
    assert L.dspfft_spec_encode(f.data_ptr(), h * w, d, gain, rangetype, scaletype, signtype, None) == 0
    gpu.cuda.synchronize()
    enc = f.cpu().numpy()
    ref = coeffs.copy()
    O = ol.lib()
    O.oracle_spec_encode_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]
    O.oracle_spec_encode_f32(ref.ctypes.data, h * w, d, gain, rangetype, scaletype, signtype)
    assert np.abs(enc - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    # decode (not invertible for abs / saturate: those drop the sign; compare with the oracle's decode of the same input)
    DC = (coeffs[0, 0].astype(np.float64)).copy()       # spec.c:66-68 stores f[z]/(4wh) of the unnormalised output == uniform-range DC
    dc_arr = (C.c_double * d)(*DC)
    assert L.dspfft_ispec_decode(f.data_ptr(), h * w, d, gain, rangetype, scaletype, signtype, dc_arr, 1, None) == 0
    gpu.cuda.synchronize()
    dec = f.cpu().numpy()
    ref2 = ref.copy()
    O.oracle_ispec_decode_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    O.oracle_ispec_decode_f32(ref2.ctypes.data, h * w, d, gain, rangetype, scaletype, signtype, DC.ctypes.data, 1)
    scale = max(1e-3, np.abs(ref2).max())
    assert np.abs(dec - ref2).max() <= 1e-5 * scale
    if signtype in (1, 3):                                # sign-preserving encodings invert back to the coefficients
        assert np.abs(dec - coeffs).max() <= 2e-5 * max(1e-3, np.abs(coeffs).max())


@pytest.mark.parametrize("cfg", [dict(damp=0.5, boost=1.0, bb=(0, 2, 3), be=(6, 20, 30)), dict(damp=1.0, boost=1.5, bb=(1, 0, 0), be=(8, 24, 40), preserve_dc=1),
                                  dict(damp=0.0, boost=2.0, bb=(0, 0, 0), be=(4, 12, 20), thr=(5.0, 5000.0), preserve_dc=2, quant=3.0),
                                  dict(damp=1.0, boost=1.0, bb=(0, 0, 0), be=(8, 24, 40), quant=7.5)])
def test_motion_filter_vs_oracle(gpu, cfg):
    import ctypes as C
    from dspfun_amd import _lib
    L = _lib.load()
    ad, ah, aw, mh, mw = 8, 24, 40, 26, 48
    c = ((ol.synth_f32(99, 10 * mh * mw) - 0.5) * 4000).astype(np.float32)
    d = dev(gpu, c)
    I3, I2 = C.c_int * 3, C.c_int * 2
    thr = cfg.get("thr", (0.0, 0.0))
    grey = 12.5
    coded = gpu.zeros(1, dtype=gpu.int64, device="cuda:0")
    assert L.dspfft_motion_filter(d.data_ptr(), I3(ad, ah, aw), I2(mh, mw), I3(*cfg["bb"]), I3(*cfg["be"]), cfg["damp"], cfg["boost"], thr[0], thr[1],
                                  cfg.get("preserve_dc", 0), grey, cfg.get("quant", 0.0), coded.data_ptr(), None) == 0
    gpu.cuda.synchronize()
    ref = c.copy()
    O = ol.lib()
    O.oracle_motion_filter_f32.restype = C.c_ulonglong
    O.oracle_motion_filter_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_double, C.c_float]
    n = O.oracle_motion_filter_f32(ref.ctypes.data, I3(ad, ah, aw), I2(mh, mw), I3(*cfg["bb"]), I3(*cfg["be"]), cfg["damp"], cfg["boost"], thr[0], thr[1],
                                   cfg.get("preserve_dc", 0), grey, cfg.get("quant", 0.0))
    assert np.array_equal(d.cpu().numpy(), ref)
    if cfg.get("quant"):
        assert int(coded[0]) == n


@pytest.mark.parametrize("case", ["volume", "frames", "frames_nofilter", "volume_quant"])
def test_fused_roundtrip_vs_unfused_and_oracle(gpu, case):
    """motion/motion.c:641-753 as one call: forward, filter, inverse with the middle axis fused into one launch;
    against the oracle's filter between separately executed plans, and against the unfused execution of the same plans"""
    import os
    from dspfun_amd import Plan, REDFT10, REDFT01
    import test_kernel_logic_cpu as tk
    r2 = float(np.sqrt(2.0))
    if case.startswith("volume"):
        d, h, w = 256, 54, 96
        n, howmany, dist, bd, active, frames = [d, h, w], 1, 0, d, (d, h, w), 1
    else:
        d, h, w = 5, 1080, 1920
        n, howmany, dist, bd, active, frames = [h, w], d, h * w, 1, (1, h, w), d
    x = ol.synth_u8(7, d * h * w).astype(np.float32).reshape(d, h, w)
    rank = len(n)
    nrm = 1.0 / np.prod([2.0 * v for v in n])
    fwd = Plan.many_r2r(n, [REDFT10] * rank, howmany=howmany, idist=dist, odist=dist).set_scale(2 * r2)
    inv = Plan.many_r2r(n, [REDFT01] * rank, howmany=howmany, idist=dist, odist=dist, first_axis_first=True).set_scale(nrm / (2 * r2))
    ref_inv = Plan.many_r2r(n, [REDFT01] * rank, howmany=howmany, idist=dist, odist=dist).set_scale(nrm / (2 * r2))
    for a in range(rank):
        fwd.set_axis_scale0(a, 1.0, 1.0 / r2); inv.set_axis_scale0(a, r2, 1.0); ref_inv.set_axis_scale0(a, r2, 1.0)
    assert "COL*" in fwd.describe().splitlines()[-1] and "COL*" in inv.describe().splitlines()[1]
    flt = None
    if case == "volume":
        flt = dict(active=active, minbuf_hw=(h, w), block_depth=bd, band_begin=(0, 1, 2), band_end=(100, 40, 70), damp=0.25, boost=1.5, preserve_dc=1)
    elif case == "volume_quant":
        flt = dict(active=active, minbuf_hw=(h, w), block_depth=bd, band_begin=(0, 0, 0), band_end=active, threshold_lo=2.0, threshold_hi=1e9, preserve_dc=2, grey_add=3.5, quantizer=4.0)
    elif case == "frames":
        flt = dict(active=active, minbuf_hw=(h, w), block_depth=bd, band_begin=(0, 3, 1), band_end=(1, 800, 1500), damp=0.5, boost=1.0, quantizer=2.5)
    dref = dev(gpu, x)
    fwd.execute(dref.data_ptr())
    gpu.cuda.synchronize()
    ref = dref.cpu().numpy()
    ncoded = 0
    if flt:
        for f in range(frames):
            ncoded += tk._oracle_filter(ref.reshape(frames, -1)[f], active, (h, w), flt)
    dref = dev(gpu, ref)
    ref_inv.execute(dref.data_ptr())
    gpu.cuda.synchronize()
    ref = dref.cpu().numpy()
    got = dev(gpu, x)
    coded = gpu.zeros(1, dtype=gpu.int64, device="cuda:0")
    fwd.roundtrip(inv, got.data_ptr(), filter=flt, d_coded=coded.data_ptr())
    gpu.cuda.synchronize()
    g = got.cpu().numpy()
    assert np.abs(g - ref).max() <= 1e-3, np.abs(g - ref).max()          # values in the u8 range; two pass orders and a quantiser in between
    if flt and flt.get("quantizer"):
        assert abs(int(coded[0]) - ncoded) <= max(2, ncoded // 100000)    # a coefficient on a rounding boundary may flip between the two orders
    if not flt:
        assert np.abs(g - x).max() < 2e-3
    os.environ["DSPFFT_NO_FUSED_ROUNDTRIP"] = "1"
    try:
        got2 = dev(gpu, x)
        coded2 = gpu.zeros(1, dtype=gpu.int64, device="cuda:0")
        fwd.roundtrip(inv, got2.data_ptr(), filter=flt, d_coded=coded2.data_ptr())
        gpu.cuda.synchronize()
    finally:
        del os.environ["DSPFFT_NO_FUSED_ROUNDTRIP"]
    assert gpu.equal(got, got2) and int(coded[0]) == int(coded2[0])        # same arithmetic, tile kept in LDS


@pytest.mark.parametrize("case", ["volume", "frames", "fallback"])
def test_roundtrip_u8_matches_float_path(gpu, case):
    """motion's 8-bit ends fused into the planar row passes (motion.c:617-640, :760-776): identical bytes to
    dspfft_u8_to_f32 -> float roundtrip -> dspfft_f32_to_u8, and within the quantiser's step of the input"""
    from dspfun_amd import Plan, _lib, REDFT10, REDFT01
    L = _lib.load()
    if case == "volume":
        d, h, w = 256, 54, 1920
        n, howmany, dist, bd, active = [d, h, w], 1, 0, d, (d, h, w)
    elif case == "frames":
        d, h, w = 4, 1080, 1920
        n, howmany, dist, bd, active = [h, w], d, h * w, 1, (1, h, w)
    else:
        d, h, w = 3, 1080, 1936
        n, howmany, dist, bd, active = [h, w], d, h * w, 1, (1, h, w)
    rank = len(n)
    u8 = ol.synth_u8(21, d * h * w)
    nrm = 1.0 / np.prod([2.0 * v for v in n])
    fwd = Plan.many_r2r(n, [REDFT10] * rank, howmany=howmany, idist=dist, odist=dist)
    inv = Plan.many_r2r(n, [REDFT01] * rank, howmany=howmany, idist=dist, odist=dist, first_axis_first=True).set_scale(nrm)
    assert ("ROW*" in fwd.describe().splitlines()[1]) == (case != "fallback"), fwd.describe()
    flt = dict(active=active, minbuf_hw=(h, w), block_depth=bd, band_begin=(0, 0, 0), band_end=active, quantizer=6.0)
    mul = 0.97
    d8 = gpu.from_numpy(u8).to("cuda:0")
    f = gpu.empty(d * h * w, dtype=gpu.float32, device="cuda:0")
    assert L.dspfft_u8_to_f32(f.data_ptr(), d8.data_ptr(), d * h * w, None) == 0
    fwd.roundtrip(inv, f.data_ptr(), filter=flt)
    ref = gpu.zeros(d * h * w, dtype=gpu.uint8, device="cuda:0")
    assert L.dspfft_f32_to_u8(ref.data_ptr(), f.data_ptr(), mul, d * h * w, None) == 0
    out = gpu.zeros(d * h * w, dtype=gpu.uint8, device="cuda:0")
    work = gpu.full((d * h * w,), float("nan"), dtype=gpu.float32, device="cuda:0")
    fwd.roundtrip_u8(inv, d8.data_ptr(), out.data_ptr(), work.data_ptr(), mul, filter=flt)
    gpu.cuda.synchronize()
    assert gpu.equal(out, ref)
    assert np.abs(out.cpu().numpy().astype(np.float64) - np.clip(np.floor(u8 * mul + 0.5), 0, 255)).max() <= 6


def test_roundtrip_of_a_plan_with_zooms_extras_runs_unfused(gpu):
    """the fused column roundtrip and the 8-bit row ends are plain instantiations (round 5): an inverse plan that carries zoom's input window on the axis
    the roundtrip fuses (its first pass), or the alternating output sign on the axis of the 8-bit end (its last pass), must take the separate passes
    and give what executing the two plans one after the other gives -- not a failed launch"""
    from dspfun_amd import Plan, _lib, REDFT10, REDFT01
    L = _lib.load()
    d, h, w = 3, 1080, 1920
    n, dist = [h, w], h * w
    nrm = 1.0 / (4.0 * h * w)
    u8 = ol.synth_u8(33, d * h * w)
    x = u8.astype(np.float32)
    for extra in ("window", "alternate"):
        fwd = Plan.many_r2r(n, [REDFT10] * 2, howmany=d, idist=dist, odist=dist)
        inv = Plan.many_r2r(n, [REDFT01] * 2, howmany=d, idist=dist, odist=dist, first_axis_first=True).set_scale(nrm)
        if extra == "window":
            assert inv.set_input_window(0, 0, h // 3) is True           # coefficient rows from h / 3 up count as zero
        else:
            assert inv.set_output_alternate(1) is True
        ref = dev(gpu, x)
        fwd.execute(ref.data_ptr()); inv.execute(ref.data_ptr())
        got = dev(gpu, x)
        fwd.roundtrip(inv, got.data_ptr())
        gpu.cuda.synchronize()
        assert gpu.equal(got, ref)
        r = ref.cpu().numpy().reshape(d, h, w)
        if extra == "alternate":
            assert np.abs(r * (1.0 - 2.0 * (np.arange(w) & 1)) - x.reshape(d, h, w)).max() < 2e-3       # the sign is there
        else:
            assert np.abs(r - x.reshape(d, h, w)).max() > 1.0                                            # ... and so is the window (a low-pass)
        d8 = gpu.from_numpy(u8).to("cuda:0")
        out = gpu.zeros(d * h * w, dtype=gpu.uint8, device="cuda:0")
        work = gpu.empty(d * h * w, dtype=gpu.float32, device="cuda:0")
        fwd.roundtrip_u8(inv, d8.data_ptr(), out.data_ptr(), work.data_ptr(), 1.0)
        want = gpu.zeros(d * h * w, dtype=gpu.uint8, device="cuda:0")
        assert L.dspfft_f32_to_u8(want.data_ptr(), ref.data_ptr(), 1.0, d * h * w, None) == 0
        gpu.cuda.synchronize()
        assert gpu.equal(out, want)


@pytest.mark.parametrize("block", [(8, 8, 8), (16, 16, 4), (5, 12, 15), (4, 4, 4), (16, 16, 16), (16, 8, 8), (8, 16, 8), (4, 16, 16)])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_guru_blocks_of_a_volume(gpu, block, dtype):
    """all blocks of a [D][H][W] volume in one plan (motion --blocksize: motion.c:591-615 walks them one by one); each block
    against the definition"""
    from dspfun_amd import Plan
    bd, bh, bw = block
    D, H, W = 4 * bd, 6 * bh, 8 * bw
    f64 = dtype == "f64"
    x = ol.synth_f32(77, D * H * W).reshape(D, H, W)
    x = x.astype(np.float64) * (1 + 2.0 ** -30) if f64 else x
    dims = [(bd, H * W, H * W), (bh, W, W), (bw, 1, 1)]
    how = [(D // bd, bd * H * W, bd * H * W), (H // bh, bh * W, bh * W), (W // bw, bw, bw)]
    for kind in (5, 4):
        p = Plan.guru(dims, how, [kind] * 3, dtype=dtype)
        assert p.describe().count("TINY") == 3, p.describe()
        # f32 blocks of 4 / 8 / 16 samples a side go through the one-pass kernel (block_core.h)
        assert ("BLOCK" in p.describe()) == (dtype == "f32" and block != (5, 12, 15)), p.describe()
        d = gpu.from_numpy(x.copy()).to("cuda:0")
        p.execute(d.data_ptr())
        gpu.cuda.synchronize()
        got = d.cpu().numpy()
        # blocks are independent: transform them as a batch with the oracle (block-major copy)
        blocks = x.reshape(D // bd, bd, H // bh, bh, W // bw, bw).transpose(0, 2, 4, 1, 3, 5).reshape(-1, bd, bh, bw).astype(np.float64)
        gb = got.reshape(D // bd, bd, H // bh, bh, W // bw, bw).transpose(0, 2, 4, 1, 3, 5).reshape(-1, bd, bh, bw)
        nb = blocks.shape[0]
        ref = ol.r2r_many(np.ascontiguousarray(blocks).ravel(), [bd, bh, bw], [kind] * 3, howmany=nb, idist=bd * bh * bw, odist=bd * bh * bw,
                          impl="port", threads=8).reshape(nb, bd, bh, bw)
        check(gb, ref, tol=1e-13 if f64 else TOL)


@pytest.mark.parametrize("seed", range(40))
def test_random_advanced_interface_geometries(gpu, seed):
    """random rank / n / howmany / stride / dist / embed, in and out of place, f32 and f64, against the definition"""
    from dspfun_amd import Plan
    import test_kernel_logic_cpu as tk
    rng = np.random.default_rng(5000 + seed)
    rank, n, howmany, embed, stride, dist, total, kinds = tk._random_case(rng)
    f64 = bool(seed % 3 == 0)
    oop = bool(rng.integers(0, 2))
    x = ol.synth_f32(seed + 1, total)
    x = x.astype(np.float64) if f64 else x
    ref_full = ol.r2r_many(x.astype(np.float64), n, kinds, howmany=howmany, inembed=embed, istride=stride, idist=dist,
                           onembed=embed, ostride=stride, odist=dist)
    p = Plan.many_r2r(n, kinds, howmany=howmany, inembed=embed, istride=stride, idist=dist, onembed=embed, ostride=stride, odist=dist,
                      dtype="f64" if f64 else "f32")
    src = gpu.from_numpy(x.copy()).to("cuda:0")
    if oop:
        out = gpu.full((total,), 7.0, dtype=src.dtype, device="cuda:0")
        p.execute(src.data_ptr(), out.data_ptr())
        gpu.cuda.synchronize()
        assert np.array_equal(src.cpu().numpy(), x)
        got, ref = out.cpu().numpy(), np.full(total, 7.0)
    else:
        p.execute(src.data_ptr())
        gpu.cuda.synchronize()
        got, ref = src.cpu().numpy(), x.astype(np.float64).copy()
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


def test_scan_pruned_idct_path(gpu):
    """scan.c:20-41,449: few coefficients per frame -> direct rank-1 sums; against the restatement and against the
    transform path (dspfft_execute_masked_accumulate) on the same frames"""
    import ctypes as C
    from dspfun_amd import _lib, REDFT10, REDFT01
    L = _lib.load()
    w, h, c = 96, 54, 3
    x = ol.synth_f32(123, w * h * c).reshape(h, w, c)
    coeffs = dev(gpu, x)
    plan_image(h, w, c, REDFT10).set_scale(1.0 / (4.0 * w * h)).execute(coeffs.data_ptr())
    gpu.cuda.synchronize()
    ch = coeffs.cpu().numpy()
    zz = ol.zigzag_order(w, h)
    step = 12                                        # 12 <= log2(96*54) = 12.3 -> the reference would prune (scan.c:349-350)
    order = dev(gpu, zz.astype(np.uint32).view(np.int32))
    total = gpu.zeros_like(coeffs); total2 = gpu.zeros_like(coeffs); work = gpu.empty_like(coeffs)
    ids = gpu.zeros(w * h, dtype=gpu.int32, device="cuda:0")
    assert L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, step, None) == 0
    inv = plan_image(h, w, c, REDFT01)
    ref = np.zeros_like(ch)
    O = ol.lib()
    O.oracle_scan_pruned_accumulate_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
    for f in range(6):
        first = f * step + (1 if f == 0 else 0)       # frame 0 skips the DC index (scan.c:445)
        cnt = (f + 1) * step - first
        assert L.dspfft_scan_pruned_accumulate(total.data_ptr(), coeffs.data_ptr(), order.data_ptr() + 4 * first, cnt, w, h, c, None) == 0
        inv.execute_masked_accumulate(coeffs.data_ptr(), work.data_ptr(), total2.data_ptr(), ids.data_ptr(), f, c)
        lin = np.ascontiguousarray(zz[first:first + cnt])
        O.oracle_scan_pruned_accumulate_f32(ref.ctypes.data, ch.ctypes.data, lin.ctypes.data, cnt, w, h, c)
        gpu.cuda.synchronize()
        assert np.abs(total.cpu().numpy() - ref).max() <= 2e-6
        assert float((total - total2).abs().max()) <= 5e-6
    # the workspace variant (no allocation inside the call): the same frames inside a captured hipGraph
    total3 = gpu.zeros_like(coeffs)
    wsf = gpu.empty(L.dspfft_scan_pruned_work_floats(step, w, h), dtype=gpu.float32, device="cuda:0")
    g = gpu.cuda.CUDAGraph()
    s_ = gpu.cuda.Stream()
    with gpu.cuda.stream(s_):
        with gpu.cuda.graph(g, stream=s_):
            for f in range(6):
                first = f * step + (1 if f == 0 else 0)
                cnt = (f + 1) * step - first
                assert L.dspfft_scan_pruned_accumulate_ws(total3.data_ptr(), coeffs.data_ptr(), order.data_ptr() + 4 * first, cnt, w, h, c, wsf.data_ptr(),
                                                          gpu.cuda.current_stream().cuda_stream) == 0
    g.replay()
    gpu.cuda.synchronize()
    assert float((total3 - total).abs().max()) == 0.0


def test_ispec_signmap_restores_the_signs_of_an_abs_spectrogram(gpu):
    """spec/ispec.c:91-99: an `abs` spectrogram plus the `saturate` sign image spec writes beside it decode back to the signed
    coefficients (every sample but the first pixel, whose map bytes carry the DC terms)"""
    from dspfun_amd import _lib
    L = _lib.load()
    h, w, d = 20, 24, 3
    c = (ol.synth_f32(8, h * w * d).reshape(h, w, d) - 0.5).astype(np.float32)
    mag = np.abs(c)
    sat = (~np.signbit(c)).astype(np.uint8) * 255                    # spec.c:134-136 as an 8-bit image
    f = dev(gpu, mag)
    m = gpu.from_numpy(sat).to("cuda:0")
    assert L.dspfft_ispec_signmap(f.data_ptr(), m.data_ptr(), h * w, d, None) == 0
    gpu.cuda.synchronize()
    got = f.cpu().numpy()
    assert np.array_equal(got.reshape(-1, d)[1:], c.reshape(-1, d)[1:])
    assert np.array_equal(got.reshape(-1, d)[0], mag.reshape(-1, d)[0])


# ---- small blocks: the one-pass kernels (block_core.h) ----
@pytest.mark.parametrize("block", [(8, 8, 8), (4, 4, 4), (16, 16, 16), (1, 8, 8), (8, 16, 4), (1, 32, 32), (1, 8, 32)])
def test_fused_block_roundtrip_gpu(gpu, block, monkeypatch):
    """motion's per-block pipeline (8-bit load, REDFT10, filter + quantiser, REDFT01, 8-bit store) in one pass: block-major stack against
    the unfused passes (DSPFFT_NO_BLOCK=1), and the blocks of a [D][H][W] volume against the same blocks rearranged block-major"""
    from dspfun_amd import Plan
    bd, bh, bw = block
    D, H, W = 4 * bd, 6 * bh, 40 * bw
    nb, vol = (D // bd) * (H // bh) * (W // bw), bd * bh * bw
    n = [v for v in (bd, bh, bw) if v > 1]
    rank = len(n)
    nrm = 1.0 / np.prod([2.0 * v for v in n])
    u8 = ol.synth_u8(11, D * H * W).reshape(D, H, W)
    flt = dict(active=(bd, bh, bw), minbuf_hw=(bh, bw), block_depth=bd, band_begin=(0, 1, 0), band_end=(bd, bh, bw - 1), damp=0.5, boost=1.25, preserve_dc=1, quantizer=3.0)
    bm = np.ascontiguousarray(u8.reshape(D // bd, bd, H // bh, bh, W // bw, bw).transpose(0, 2, 4, 1, 3, 5))
    res = {}
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("DSPFFT_NO_BLOCK", raising=False)
        else:
            monkeypatch.setenv("DSPFFT_NO_BLOCK", "1")
        fb = Plan.many_r2r(n, [5] * rank, howmany=nb, idist=vol, odist=vol)
        ib = Plan.many_r2r(n, [4] * rank, howmany=nb, idist=vol, odist=vol, first_axis_first=True).set_scale(nrm)
        assert ("BLOCK" in fb.describe()) == fused
        din = gpu.from_numpy(bm.copy()).to("cuda:0"); dout = gpu.zeros_like(din); work = gpu.empty(nb * vol, dtype=gpu.float32, device="cuda:0")
        coded = gpu.zeros(1, dtype=gpu.int64, device="cuda:0")
        fb.roundtrip_u8(ib, din.data_ptr(), dout.data_ptr(), work.data_ptr(), 1.0, filter=flt, d_coded=coded.data_ptr())
        gpu.cuda.synchronize()
        res[fused] = (dout.cpu().numpy(), int(coded.item()))
    a, b = res[True], res[False]
    assert np.abs(a[0].astype(np.int32) - b[0].astype(np.int32)).max() <= 1 and (a[0] != b[0]).mean() < 1e-3
    assert abs(a[1] - b[1]) <= max(4, b[1] // 10000) and b[1] > 0
    monkeypatch.delenv("DSPFFT_NO_BLOCK", raising=False)
    dims = [d for d in [(bd, H * W, H * W), (bh, W, W), (bw, 1, 1)] if d[0] > 1]
    how = [(D // bd, bd * H * W, bd * H * W), (H // bh, bh * W, bh * W), (W // bw, bw, bw)]
    fv = Plan.guru(dims, how, [5] * rank)
    iv = Plan.guru(dims, how, [4] * rank).set_scale(nrm)
    assert "side by side" in fv.describe()
    din = gpu.from_numpy(u8.copy()).to("cuda:0"); dout = gpu.zeros_like(din); work = gpu.empty(D * H * W, dtype=gpu.float32, device="cuda:0")
    coded = gpu.zeros(1, dtype=gpu.int64, device="cuda:0")
    fv.roundtrip_u8(iv, din.data_ptr(), dout.data_ptr(), work.data_ptr(), 1.0, filter=flt, d_coded=coded.data_ptr())
    gpu.cuda.synchronize()
    back = a[0].reshape(D // bd, H // bh, W // bw, bd, bh, bw).transpose(0, 3, 1, 4, 2, 5).reshape(D, H, W)
    assert np.array_equal(dout.cpu().numpy(), back) and int(coded.item()) == a[1]


# ---- plan-time specialisation (DSPFFT_JIT=1): frame sizes without an entry in spec_list.h get RowSpecT / ColSpecT kernels compiled with hiprtc ----
@pytest.mark.parametrize("h,w,c,dtype", [(1000, 1500, 3, "f32"), (750, 1000, 3, "f32"), (1500, 2000, 1, "f32"), (1350, 2400, 3, "f32"), (750, 1000, 3, "f64")])
def test_plan_time_specialisation(gpu, h, w, c, dtype, monkeypatch):
    from dspfun_amd import Plan, REDFT10, REDFT01
    monkeypatch.setenv("DSPFFT_JIT", "1")
    f64 = dtype == "f64"
    x = ol.synth_f32(h + 3 * w, h * w * c).reshape(h, w, c)
    x = x.astype(np.float64) * (1 + 2.0 ** -31) if f64 else x
    fwd = Plan.image(h, w, c, REDFT10, dtype=dtype)
    inv = Plan.image(h, w, c, REDFT01, dtype=dtype).set_scale(1.0 / (4.0 * h * w))
    assert fwd.describe().count("compiled at plan time") == 2 and inv.describe().count("compiled at plan time") == 2, fwd.describe()
    d = gpu.from_numpy(x.copy()).to("cuda:0")
    o = gpu.empty_like(d)
    fwd.execute(d.data_ptr(), o.data_ptr())
    gpu.cuda.synchronize()
    ref = ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=8)
    check(o.cpu().numpy(), ref, tol=1e-13 if f64 else TOL)
    inv.execute(o.data_ptr())
    gpu.cuda.synchronize()
    assert np.abs(o.cpu().numpy() - x).max() <= (1e-13 if f64 else 2e-6)
    # the fused scan step runs on the compiled kernels too (masked loads, accumulating stores)
    if not f64:
        from dspfun_amd import _lib
        L = _lib.load()
        coeffs = gpu.from_numpy(x.copy()).to("cuda:0")
        Plan.image(h, w, c, REDFT10).set_scale(1.0 / (4.0 * h * w)).execute(coeffs.data_ptr())
        ids = gpu.zeros(h * w, dtype=gpu.int32, device="cuda:0")
        assert L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, (h * w + 4) // 5, None) == 0
        acc = gpu.empty_like(coeffs); work = gpu.empty_like(coeffs)
        assert L.dspfft_broadcast_dc(acc.data_ptr(), coeffs.data_ptr(), w * h, c, None) == 0
        p01 = Plan.image(h, w, c, REDFT01)
        for f in range(5):
            p01.execute_masked_accumulate(coeffs.data_ptr(), work.data_ptr(), acc.data_ptr(), ids.data_ptr(), f, c)
        gpu.cuda.synchronize()
        assert float((acc.cpu() - gpu.from_numpy(x)).abs().max()) <= 5e-6


def test_plan_effort_and_disk_cache(gpu, monkeypatch, tmp_path):
    """dspfft_set_plan_effort(1) compiles the kernels of an unlisted size at plan time and leaves the code objects in the disk cache
    ($DSPFFT_JIT_CACHE; a later process loads them in about a millisecond instead of compiling for a second or two); effort 0 keeps the
    runtime-geometry kernels; DSPFFT_JIT=2 overrides"""
    from dspfun_amd import Plan, _lib, REDFT10
    L = _lib.load()
    monkeypatch.delenv("DSPFFT_JIT", raising=False)
    monkeypatch.setenv("DSPFFT_JIT_CACHE", str(tmp_path / "jit"))
    h, w, c = 700, 900, 3
    assert "compiled at plan time" not in Plan.image(h, w, c, REDFT10).describe()
    L.dspfft_set_plan_effort(1)
    try:
        p = Plan.image(h, w, c, REDFT10)
        assert p.describe().count("compiled at plan time") == 2, p.describe()
        files = sorted(os.listdir(tmp_path / "jit"))
        assert len(files) == 2 and all(f.endswith(".co") for f in files)
        monkeypatch.setenv("DSPFFT_JIT", "2")
        assert "compiled at plan time" not in Plan.image(h, w, c, REDFT10).describe()
    finally:
        L.dspfft_set_plan_effort(0)
    x = ol.synth_f32(5, h * w * c).reshape(h, w, c)
    d = gpu.from_numpy(x.copy()).to("cuda:0")
    p.execute(d.data_ptr())
    gpu.cuda.synchronize()
    check(d.cpu().numpy(), ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=8))


def test_plan_effort_2_times_candidates(gpu, monkeypatch, tmp_path):
    """planning effort 2 (FFTW_PATIENT / FFTW_EXHAUSTIVE through the shim): several candidate kernels are compiled and timed on a scratch
    buffer; the plan that comes out is a compiled one and correct, and the caller's arrays were never involved"""
    from dspfun_amd import Plan, set_plan_effort, REDFT10, REDFT01
    monkeypatch.delenv("DSPFFT_JIT", raising=False)
    monkeypatch.delenv("DSPFFT_JIT_TUNE", raising=False)
    monkeypatch.setenv("DSPFFT_JIT_CACHE", str(tmp_path / "jit"))
    h, w, c = 600, 1100, 3
    set_plan_effort(2)
    try:
        fwd = Plan.image(h, w, c, REDFT10)
        inv = Plan.image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * h * w))
    finally:
        set_plan_effort(0)
    assert fwd.describe().count("compiled at plan time") == 2, fwd.describe()
    assert len(os.listdir(tmp_path / "jit")) >= 4          # more than one candidate per pass went through the compiler
    x = ol.synth_f32(8, h * w * c).reshape(h, w, c)
    d = gpu.from_numpy(x.copy()).to("cuda:0")
    fwd.execute(d.data_ptr())
    gpu.cuda.synchronize()
    check(d.cpu().numpy(), ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=8))
    inv.execute(d.data_ptr())
    gpu.cuda.synchronize()
    assert np.abs(d.cpu().numpy() - x).max() <= 2e-6


def test_plan_time_kernels_run_motions_fused_pipeline(gpu, monkeypatch, tmp_path):
    """a frame size without listed kernels (1000x600 luma, 12 frames): with kernels compiled at plan time motion's per-frame pipeline
    -- 8-bit in, REDFT10, quantiser, REDFT01, 8-bit out -- runs as three launches (8-bit row kernels + fused column roundtrip) and
    matches the runtime-geometry path (DSPFFT_JIT=2)"""
    from dspfun_amd import Plan, REDFT10, REDFT01
    monkeypatch.setenv("DSPFFT_JIT_CACHE", str(tmp_path / "jit"))
    d_, h, w = 12, 600, 1000
    u8 = ol.synth_u8(3, d_ * h * w).reshape(d_, h, w)
    flt = dict(active=(1, h, w), minbuf_hw=(h, w), block_depth=1, band_begin=(0, 0, 0), band_end=(1, h, w), quantizer=4.0)
    res = {}
    for jit in ("1", "2"):
        monkeypatch.setenv("DSPFFT_JIT", jit)
        f2 = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=d_, idist=h * w, odist=h * w)
        i2 = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=d_, idist=h * w, odist=h * w, first_axis_first=True).set_scale(1.0 / (4.0 * h * w))
        assert (f2.describe().count("compiled at plan time") == 2) == (jit == "1"), f2.describe()
        din = gpu.from_numpy(u8.copy()).to("cuda:0"); dout = gpu.zeros_like(din); work = gpu.empty(d_ * h * w, dtype=gpu.float32, device="cuda:0")
        coded = gpu.zeros(1, dtype=gpu.int64, device="cuda:0")
        f2.roundtrip_u8(i2, din.data_ptr(), dout.data_ptr(), work.data_ptr(), 1.0, filter=flt, d_coded=coded.data_ptr())
        gpu.cuda.synchronize()
        res[jit] = (dout.cpu().numpy(), int(coded.item()))
    a, b = res["1"], res["2"]
    assert np.abs(a[0].astype(np.int32) - b[0].astype(np.int32)).max() <= 1 and (a[0] != b[0]).mean() < 1e-3
    assert abs(a[1] - b[1]) <= max(4, b[1] // 10000) and b[1] > 0
    assert np.abs(a[0].astype(np.int32) - u8).max() <= 6          # quantiser 4: a coarse but close copy


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_huge_prime_length_gets_a_working_plan(gpu, dtype):
    """the reference never checks for a NULL plan (spec.c:63-64).  N = 50021 is prime and beyond the LDS range of both the
    Bluestein and the dense pass: the line is staged in device memory (engine.cpp DENSE).  Selected coefficients against the definition
    in f64, and the 2N roundtrip identity."""
    from dspfun_amd import Plan, REDFT10, REDFT01
    N = 50021
    npd = np.float32 if dtype == "f32" else np.float64
    x = (ol.synth_f32(N, N).astype(np.float64) - 0.5).astype(npd)
    fwd = Plan.many_r2r([N], [REDFT10], dtype=dtype)
    assert "staged in device memory" in fwd.describe(), fwd.describe()
    d = dev(gpu, x)
    fwd.execute(d.data_ptr())
    gpu.cuda.synchronize()
    got = d.cpu().numpy().astype(np.float64)
    j = np.arange(N)
    scale = np.abs(got).max()
    tol = 2e-5 if dtype == "f32" else 1e-11
    for k in (0, 1, 2, 777, 25010, 50020):
        want = 2.0 * np.sum(x.astype(np.float64) * np.cos(np.pi * (j + 0.5) * k / N))
        assert abs(got[k] - want) <= tol * scale, (k, got[k], want)
    Plan.many_r2r([N], [REDFT01], dtype=dtype).set_scale(1.0 / (2.0 * N)).execute(d.data_ptr())
    gpu.cuda.synchronize()
    assert np.abs(d.cpu().numpy().astype(np.float64) - x).max() <= (5e-5 if dtype == "f32" else 1e-11)


def test_execute_many_repeat_on_two_library_streams(gpu):
    """bench.py's step loop (dspfft_execute_many_repeat) on real streams: four frames on two library-owned streams, seven repeats with a
    re-join every two, per-pass event windows rotating through the frames; afterwards every frame is the input again and the events
    carry positive times"""
    from dspfun_amd import Plan, REDFT10, REDFT01
    from dspfun_amd.engine import Batch, Events, Stream
    h, w, c = 540, 960, 3
    fwd = Plan.image(h, w, c, REDFT10)
    inv = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=c, istride=c, idist=1, ostride=c, odist=1, first_axis_first=True).set_scale(1.0 / (4.0 * w * h))
    x = ol.synth_f32(0xD5F0A01, 4 * h * w * c).reshape(4, h, w, c)
    d = dev(gpu, x)
    streams = [Stream(), Stream()]
    batch = Batch([(pl, d[f].data_ptr(), None, streams[f % 2].handle) for f in range(4) for pl in (fwd, inv)])
    npass = fwd.num_passes + inv.num_passes
    reps, every, window = 7, 3, 4                                   # windows on repeats 0, 3, 6
    ev = Events(2 * npass * 2 * 3)
    gpu.cuda.synchronize()
    batch.run_repeat(reps, 2, every, window, ev)
    for s_ in streams:
        s_.synchronize()
    assert np.abs(d.cpu().numpy() - x).max() <= 3e-5               # seven in-place roundtrips
    times = [ev.elapsed_ms(2 * j, 2 * j + 1) for j in range(npass * 2 * 3)]
    assert all(0.0 < t < 50.0 for t in times), times
