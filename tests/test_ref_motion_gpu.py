"""GPU (-m gpu): motion's elementwise kernels against the reference's own compiled lines (tests/golden/ref_motion.npz, see
tests/test_ref_motion.py): dspfft_motion_filter (motion/motion.c:683-744: six-face damp / boost, threshold, DC preservation, quantiser and
its count of coded coefficients) and dspfft_f32_to_u8 (:759-776: scale, clamp, lround) through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from test_ref_motion import FIX, NCASES, case, check_filtered, I2, I3

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
