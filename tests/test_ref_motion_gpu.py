"""GPU (-m gpu): motion's elementwise kernels against the reference's own compiled lines (tests/golden/ref_motion.npz, see
tests/test_ref_motion.py): dspfft_motion_filter (motion/motion.c:683-744: six-face damp / boost, threshold, DC preservation, quantiser and
its count of coded coefficients) and dspfft_f32_to_u8 (:759-776: scale, clamp, lround) through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from test_ref_motion import FIX, NCASES, case, check_filtered, I2, I3, IO_GEOMS, ISPEC_MODES, SPEC_MODES, io_consts, check_load, check_store

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dspfun_amd import _lib
    return torch, _lib.load()


@pytest.mark.parametrize("ci", range(NCASES))
def test_motion_filter_kernel_is_the_references(gpu, ci):
    torch, L = gpu
    p = case(ci)
    buf = torch.from_numpy(FIX[f"m{ci}_in"].copy()).to("cuda:0")
    coded = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    rc = L.dspfft_motion_filter(buf.data_ptr(), I3(*p["active"]), I2(p["minbuf"][1], p["minbuf"][2]), I3(*p["band_begin"]), I3(*p["band_end"]), p["damp"], p["boost"],
                                p["threshold_lo"], p["threshold_hi"], p["preserve_dc"], p["grey_add"], p["quantizer"], coded.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    check_filtered(buf.cpu().numpy(), int(coded.item()), p, ci)


@pytest.mark.parametrize("ci", range(NCASES))
def test_f32_to_u8_kernel_is_the_references(gpu, ci):
    torch, L = gpu
    p = case(ci)
    d, h, w = p["active"]
    md, mh, mw = p["minbuf"]
    src = torch.from_numpy(FIX[f"m{ci}_store_in"].copy()).to("cuda:0")
    dst = torch.zeros(src.numel(), dtype=torch.uint8, device="cuda:0")
    assert L.dspfft_f32_to_u8(dst.data_ptr(), src.data_ptr(), 1.0 / (8.0 * w * h * d), src.numel(), None) == 0
    torch.cuda.synchronize()
    g = dst.cpu().numpy().reshape(md, mh, mw)[:d, :h, :w].ravel()
    r = FIX[f"m{ci}_store_u8"].reshape(md, mh, mw)[:d, :h, :w].ravel()
    assert np.array_equal(g[16:], r[16:])            # (the first 16 are constructed ties: see tests/test_ref_motion.py)
    assert np.abs(g[:16].astype(int) - r[:16].astype(int)).max() <= 1


@pytest.mark.parametrize("g", range(len(IO_GEOMS)))
@pytest.mark.parametrize("px", ["u8", "f32"])
@pytest.mark.parametrize("name", list(ISPEC_MODES))
def test_motion_load_kernels_are_the_references(gpu, g, px, name):
    """dspfft_motion_load_u8 / _f32 against motion.c:617-638 compiled as they lie: every --ispec decode, 8-bit and float pixels"""
    torch, L = gpu
    d, h, w, md, mh, mw = IO_GEOMS[g]
    _, norm, _, ic = io_consts(g)
    pix = torch.from_numpy(FIX[f"io{g}_pix_{px}"].copy()).to("cuda:0")
    c = torch.zeros(md * mh * mw, dtype=torch.float32, device="cuda:0")
    fn = L.dspfft_motion_load_f32 if px == "f32" else L.dspfft_motion_load_u8
    assert fn(c.data_ptr(), pix.data_ptr(), I3(d, h, w), I2(mh, mw), ISPEC_MODES[name], ic, norm, None) == 0, L.dspfft_motion_last_error()
    torch.cuda.synchronize()
    check_load(c.cpu().numpy(), g, name, px)


@pytest.mark.parametrize("g", range(len(IO_GEOMS)))
@pytest.mark.parametrize("px", ["u8", "f32"])
@pytest.mark.parametrize("name", list(SPEC_MODES))
def test_motion_store_kernels_are_the_references(gpu, g, px, name):
    """dspfft_motion_store_u8 / _f32 against motion.c:755-776 compiled as they lie: every --spec encode, 8-bit and float pixels"""
    torch, L = gpu
    d, h, w, md, mh, mw = IO_GEOMS[g]
    sf, norm, cc, _ = io_consts(g)
    c = torch.from_numpy(FIX[f"io{g}_coeffs"].copy()).to("cuda:0")
    out = torch.zeros(md * mh * mw, dtype=torch.float32 if px == "f32" else torch.uint8, device="cuda:0")
    fn = L.dspfft_motion_store_f32 if px == "f32" else L.dspfft_motion_store_u8
    assert fn(out.data_ptr(), c.data_ptr(), I3(d, h, w), I2(mh, mw), SPEC_MODES[name], sf, norm, cc.get(name, 0.0), None) == 0, L.dspfft_motion_last_error()
    torch.cuda.synchronize()
    check_store(out.cpu().numpy(), g, name, px)
