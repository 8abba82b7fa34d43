"""GPU (-m gpu): BASELINE config 3 at `north_star`'s tolerance over the WHOLE frame -- max|gpu - ref| <= 1e-5 max|ref| -- for every zoom path
(fast transforms on the DCT-III grid, chirp-z, dense MFMA product), and the 3.7x `centered` frame for the two paths that take it.
The reference frame is the f64 restatement of zoom/zoom.c:36-68 (basis, from the oracle) and :361-375 (the separable product with the DC
row / column halved, divided by w h), the two products run as f64 matrix products.  The measured errors are printed and, when
gpurun_out/ exists, written to gpurun_out/zoom_c3_error.json (copied to profiles/ by hand)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dspfun_amd import _lib
    _lib.load()
    return torch


def reference_frame(gpu, x, btype, scale, vx, vy, vw, vh):
    """(vh, vw, 3) f64 on the device: out = YB (C XB^T) / (w h), DC column / row of the bases = 1/2 (zoom.c:361-375)"""
    h, w, _ = x.shape
    L = ol.lib()
    cf = ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10, impl="port", threads=8)

    def basis(off, nv, n):
        nc = L.oracle_zoom_basis_f64(None, btype, scale, 1.0, off, nv, n)
        b = np.zeros(nv * (nc - 1))
        L.oracle_zoom_basis_f64(b.ctypes.data, btype, scale, 1.0, off, nv, n)
        return np.concatenate([np.full((nv, 1), 0.5), b.reshape(nv, nc - 1)], axis=1), nc
    XB, cw = basis(vx, vw, w)
    YB, ch = basis(vy, vh, h)
    d = "cuda:0"
    C_ = gpu.from_numpy(np.ascontiguousarray(cf[:ch, :cw])).to(d)                   # (ch, cw, 3)
    T = gpu.from_numpy(YB).to(d) @ C_.reshape(ch, cw * 3)                          # (vh, cw * 3)
    T = T.reshape(vh, cw, 3).permute(1, 0, 2).reshape(cw, vh * 3)
    out = gpu.from_numpy(XB).to(d) @ T                                              # (vw, vh * 3)
    return (out.reshape(vw, vh, 3).permute(1, 0, 2) / (w * h)).contiguous()


RESULTS = {}


def record(name, got, ref):
    err = float((got.double() - ref).abs().max())
    mx = float(ref.abs().max())
    RESULTS[name] = {"max_abs_err": err, "max_abs_ref": mx, "rel": err / mx}
    print(f"\n{name}: max|gpu - ref| = {err:.3e}, max|ref| = {mx:.4f}, ratio = {err / mx:.3e}")
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        with open(os.path.join(ROOT, "gpurun_out", "zoom_c3_error.json"), "w") as f:
            json.dump(RESULTS, f, indent=1)
    return err, mx


@pytest.mark.parametrize("method", ["fft", "czt", "gemm"])
def test_c3_whole_frame_at_the_stated_tolerance(gpu, method):
    from dspfun_amd.zoom import Zoom
    w, h = 1920, 1080
    x = ol.synth_f32(0xD5F0003, w * h * 3).reshape(h, w, 3)
    z = Zoom(gpu, gpu.from_numpy(x).to("cuda:0"))
    got = z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), method=method)
    ref = reference_frame(gpu, x, 0, 4.0, 0.0, 0.0, 4 * w, 4 * h)
    err, mx = record(f"c3_4x_interpolated_{method}", got, ref)
    assert err <= 1e-5 * mx
    # the samples coincide with the input at integer scale
    assert float((got[::4, ::4] - gpu.from_numpy(x).to("cuda:0")).abs().max()) <= 1e-5 * mx


@pytest.mark.parametrize("method", ["czt", "gemm"])
def test_1080p_3p7_centered_whole_frame_at_the_stated_tolerance(gpu, method):
    from dspfun_amd.zoom import Zoom
    w, h = 1920, 1080
    x = ol.synth_f32(0xD5F0003, w * h * 3).reshape(h, w, 3)
    z = Zoom(gpu, gpu.from_numpy(x).to("cuda:0"))
    vw, vh = int(w * 3.7), int(h * 3.7)
    got = z.frame(vw, vh, (3.7, 1.0), (3.7, 1.0), 12.5, -4.25, 1, method=method)
    ref = reference_frame(gpu, x, 1, 3.7, 12.5, -4.25, vw, vh)
    err, mx = record(f"1080p_3p7_centered_{method}", got, ref)
    assert err <= 1e-5 * mx
