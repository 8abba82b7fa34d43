"""GPU (-m gpu): dspfft_cosrows_* -- zoom's x stage (zoom/zoom.c:361-368 on a DCT-III grid) on the duo row kernel (dct_duo.h): every
listed scaled length against the cosine series in float64, every source-pixel count, clipped viewports and pitches; BASELINE config 3's
lines against the two-transform row pass it replaces (dspfft_execute_sum2 through dspfft_zoomfft_*, DSPFFT_ZOOM_XROWS=0)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

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


def series(x, M, vw, theta, scale, rows):
    cw = x.shape[1]
    u = np.arange(cw)[None, :]
    b = np.arange(vw)[:, None]
    basis = np.cos(u * (np.pi * (b + 0.5) / M + theta))
    basis[:, 0] *= 0.5
    return scale * np.einsum("bu,juc->jbc", basis, x[rows].astype(np.float64))


def cosrows(gpu, x, M, vw, theta, scale, in_pad=0, out_pad=0):
    from dspfun_amd import _lib
    L = _lib.load()
    lines, cw, _ = x.shape
    ip, op = cw * 3 + in_pad, vw * 3 + out_pad
    src = gpu.zeros(lines * ip, dtype=gpu.float32, device="cuda:0")
    src.view(lines, ip)[:, :cw * 3] = gpu.from_numpy(x.reshape(lines, cw * 3)).to("cuda:0")
    dst = gpu.full((lines * op,), -77.0, dtype=gpu.float32, device="cuda:0")
    p = C.c_void_p()
    assert L.dspfft_cosrows_create(C.byref(p), M, cw, vw, lines) == 0, L.dspfft_last_error()
    try:
        assert L.dspfft_cosrows_execute(p, src.data_ptr(), ip, dst.data_ptr(), op, theta, scale, None) == 0, L.dspfft_last_error()
        gpu.cuda.synchronize()
    finally:
        L.dspfft_cosrows_destroy(p)
    d = dst.view(lines, op).cpu().numpy()
    return d[:, :vw * 3].reshape(lines, vw, 3), d[:, vw * 3:]


@pytest.mark.parametrize("M,cw,vw,theta", [
    (7680, 1920, 7680, 0.00417),         # BASELINE config 3's shape, panned
    (7680, 1920, 7680, 0.0),
    (7680, 1000, 5000, -0.2),            # clipped viewport, fewer coefficients
    (7680, 3840, 7680, 0.001),           # two source pixels per slot pair (scale 2)
    (7680, 7680, 7679, 0.003),           # four (scale 1)
    (5760, 1920, 5760, 0.002),
    (5760, 1440, 4000, 0.1),
    (3840, 1920, 3840, 0.01),
    (3840, 960, 3840, -0.01),
    (2560, 640, 2560, 0.02),
    (2560, 1280, 2000, 0.0),
    (1920, 480, 1920, 0.002),
    (1280, 320, 1280, 0.3),
    (1280, 1, 1280, 0.5),
])
def test_cosine_series_rows(gpu, M, cw, vw, theta):
    rng = np.random.default_rng(M + cw)
    lines = 37
    x = rng.random((lines, cw, 3), dtype=np.float32) - np.float32(0.5)
    got, _ = cosrows(gpu, x, M, vw, theta, 1.0 / 3.0)
    rows = [0, 1, 17, lines - 1]
    ref = series(x, M, vw, theta, 1.0 / 3.0, rows)
    assert np.abs(got[rows] - ref).max() <= 1e-5 * np.abs(ref).max()
    # every line is some line's result, not garbage: lines are independent, so equal inputs give equal outputs
    y = np.repeat(x[5:6], 3, axis=0)
    g2, _ = cosrows(gpu, y, M, vw, theta, 1.0 / 3.0)
    assert np.array_equal(g2[0], g2[1]) and np.array_equal(g2[0], g2[2]) and np.array_equal(g2[0], got[5])


def test_pitches_leave_the_gaps_alone(gpu):
    M, cw, vw = 3840, 960, 3000
    x = np.random.default_rng(5).random((9, cw, 3), dtype=np.float32)
    got, pad = cosrows(gpu, x, M, vw, 0.01, 2.0, in_pad=7, out_pad=5)
    ref = series(x, M, vw, 0.01, 2.0, list(range(9)))
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.all(pad == np.float32(-77))


def test_c3_frame_equals_the_two_transform_row_pass(gpu, tmp_path):
    """BASELINE config 3 through dspfft_zoomfft_* with the duo row kernel (default) and with the row pass of round 3
    (DSPFFT_ZOOM_XROWS=0, a child process: the switch is read at plan time): same frame to rounding, panned and clipped."""
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle_lib as ol
from dspfun_amd.zoom import Zoom
w, h = 1920, 1080
x = ol.synth_f32(0xD5F0003, w * h * 3).reshape(h, w, 3)
z = Zoom(torch, torch.from_numpy(x).to("cuda:0"))
a = z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), vx=100.25, vy=50.5, method="fft")
b = z.frame(3000, 2000, (4.0, 1.0), (4.0, 1.0), vx=7.0, vy=3.0, method="fft")
torch.cuda.synchronize()
np.save(sys.argv[1], a[::7, ::5].cpu().numpy()); np.save(sys.argv[2], b[::3, ::3].cpu().numpy())
''' % (os.path.dirname(HERE), HERE)
    outs = {}
    for tag, env in (("duo", {}), ("sum2", {"DSPFFT_ZOOM_XROWS": "0"})):
        e = dict(os.environ); e.update(env)
        fa, fb = str(tmp_path / f"{tag}_a.npy"), str(tmp_path / f"{tag}_b.npy")
        subprocess.check_call([sys.executable, "-c", code, fa, fb], env=e)
        outs[tag] = (np.load(fa), np.load(fb))
    for i in range(2):
        assert np.abs(outs["duo"][i] - outs["sum2"][i]).max() <= 2e-5 * max(1.0, np.abs(outs["sum2"][i]).max())
