"""CPU (-m "not gpu"): the phases of the duo row kernel behind dspfft_cosrows_* (dspfun_amd/csrc/dct_duo.h: zoom's x stage, zoom/zoom.c:361-368
on a DCT-III grid) through the test-only emulation, against the cosine series itself evaluated in float64 -- every source-pixel count
(cw <= M/4, <= M/2, <= M), clipped viewports, line pitches, pans."""
import ctypes as C

import numpy as np
import pytest

from emul_lib import emul


def series(x, M, vw, theta, scale):
    """out[j][b][c] = scale * sum'_u x[j][u][c] cos(u (pi (b + 1/2) / M + theta)) in float64"""
    cw = x.shape[1]
    u = np.arange(cw)[None, :]
    b = np.arange(vw)[:, None]
    basis = np.cos(u * (np.pi * (b + 0.5) / M + theta))
    basis[:, 0] *= 0.5
    return scale * np.einsum("bu,juc->jbc", basis, x.astype(np.float64))


def run(L, x, M, vw, theta, scale, in_pitch=None, out_pitch=None):
    lines, cw, _ = x.shape
    in_pitch = in_pitch or cw * 3
    out_pitch = out_pitch or vw * 3
    src = np.zeros(lines * in_pitch, dtype=np.float32)
    for j in range(lines):
        src[j * in_pitch:j * in_pitch + cw * 3] = x[j].ravel()
    dst = np.full(lines * out_pitch, np.float32(-77))
    p = C.c_void_p()
    assert L.dspfft_cosrows_create(C.byref(p), M, cw, vw, lines) == 0, L.dspfft_last_error()
    try:
        assert L.dspfft_cosrows_execute(p, src.ctypes.data, in_pitch, dst.ctypes.data, out_pitch, theta, scale, None) == 0, L.dspfft_last_error()
    finally:
        L.dspfft_cosrows_destroy(p)
    out = np.stack([dst[j * out_pitch:j * out_pitch + vw * 3].reshape(vw, 3) for j in range(lines)])
    pad = np.concatenate([dst[j * out_pitch + vw * 3:(j + 1) * out_pitch] for j in range(lines)]) if out_pitch > vw * 3 else np.zeros(0)
    return out, pad


@pytest.mark.parametrize("M,cw,vw,theta", [
    (1280, 320, 1280, 0.0),              # one source pixel per slot pair (scale 4), no pan: 1/2 REDFT01 of the zero-padded line
    (1280, 320, 1280, 0.0123),
    (1280, 200, 1000, -0.4),             # clipped viewport
    (1280, 640, 1280, 0.0123),           # two sources (scale 2)
    (1280, 500, 1279, 0.3),
    (1280, 1280, 1280, 0.0123),          # four (scale 1)
    (1280, 900, 640, 0.07),
    (1920, 480, 1920, 0.002),
    (1920, 1, 1920, 0.5),                # the constant term alone
])
def test_cosine_series_rows(M, cw, vw, theta):
    L = emul()
    rng = np.random.default_rng(M + cw)
    x = (rng.random((3, cw, 3), dtype=np.float32) - np.float32(0.5))
    got, _ = run(L, x, M, vw, theta, 1.0 / 777.0)
    ref = series(x, M, vw, theta, 1.0 / 777.0)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_pitches_leave_the_gaps_alone():
    L = emul()
    M, cw, vw = 1280, 320, 1100
    x = (np.random.default_rng(5).random((4, cw, 3), dtype=np.float32))
    got, pad = run(L, x, M, vw, 0.01, 2.0, in_pitch=cw * 3 + 7, out_pitch=vw * 3 + 5)
    ref = series(x, M, vw, 0.01, 2.0)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.all(pad == np.float32(-77))


def test_theta_zero_is_half_redft01_of_the_padded_line():
    import oracle_lib as ol
    L = emul()
    M, cw = 1280, 320
    x = np.random.default_rng(9).random((1, cw, 3), dtype=np.float32)
    got, _ = run(L, x, M, M, 0.0, 1.0)
    z = np.zeros((M, 3))
    z[:cw] = x[0]
    for c in range(3):
        ref = 0.5 * ol.r2r_many(z[:, c].copy(), [M], [ol.REDFT01])
        assert np.abs(got[0, :, c] - ref).max() <= 1e-5 * np.abs(ref).max()


def test_refusals():
    L = emul()
    p = C.c_void_p()
    assert L.dspfft_cosrows_create(C.byref(p), 1000, 250, 1000, 1) == -2          # no listed kernel
    assert L.dspfft_cosrows_create(C.byref(p), 1280, 1281, 1280, 1) == -1         # more coefficients than samples
    assert L.dspfft_cosrows_create(C.byref(p), 1280, 320, 1281, 1) == -1          # viewport wider than the line
