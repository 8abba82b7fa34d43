"""CPU (-m "not gpu"): the phases of the chirp-z row kernel behind dspfft_cztrows_* (dspfun_amd/csrc/dct_czt.h: zoom's product at ANY sample
spacing, zoom/zoom.c:36-68,361-375) through the test-only emulation, against the cosine series itself in float64 -- the three bases'
(omega, phi) at integer and non-integer scales, strided and grouped lines, every listed stage count."""
import ctypes as C

import numpy as np
import pytest

from emul_lib import emul


def series(x, nout, omega, phi, scale):
    """out[l][b] = scale * sum'_n x[l][n] cos(n (omega b + phi)) in float64"""
    nc = x.shape[-1]
    n = np.arange(nc)[None, :]
    b = np.arange(nout)[:, None]
    basis = np.cos(n * (omega * b + phi))
    basis[:, 0] *= 0.5
    return scale * (x.astype(np.float64) @ basis.T)


def axis(typ, num, den, length, off):
    """(omega, phi) of zoom.c:49-61"""
    if typ == 2:
        alpha, N = 1.0, length * num / den
    elif typ == 0:
        alpha, N = den / num, float(length)
    else:
        alpha, N = (length - 1) * den / (length * num - den), float(length)
    return np.pi * alpha / N, np.pi * (alpha * off + 0.5) / N


def run(L, x, nout, omega, phi, scale, group=1):
    lines, nc = x.shape
    src = np.ascontiguousarray(x, dtype=np.float32)
    dst = np.full((lines, nout), np.float32(-77))
    p = C.c_void_p()
    assert L.dspfft_cztrows_create(C.byref(p), nc, nout, lines, 1) == 0, L.dspfft_last_error()
    try:
        assert L.dspfft_cztrows_execute(p, src.ctypes.data, nc, 0, 1, dst.ctypes.data, nout, 0, 1, omega, phi, scale, None) == 0, L.dspfft_last_error()
        P = L.dspfft_cztrows_length(p)
    finally:
        L.dspfft_cztrows_destroy(p)
    return dst, P


@pytest.mark.parametrize("nc,nout,typ,num,den,off", [
    (300, 900, 0, 3.0, 1.0, 0.0),            # P = 1200 (three stages), integer scale: the DCT-III grid as a special case
    (300, 850, 0, 2.83, 1.0, 4.25),          # non-integer scale, panned
    (300, 850, 1, 2.83, 1.0, 4.25),          # centered
    (300, 850, 2, 2.83, 1.0, -3.5),          # native
    (640, 1700, 1, 3.7, 1.3, 10.0),          # P = 2400
    (1080, 4320, 1, 4.0, 1.0, 0.0),          # P = 5400 (four stages): config 3's y axis, centered
    (1000, 700, 0, 0.7, 1.0, 0.0),           # down-scale: fewer samples than coefficients (nc = round(len * scale))
    (1, 50, 0, 1.0, 1.0, 0.0),               # the constant term alone
])
def test_cosine_series_rows_any_spacing(nc, nout, typ, num, den, off):
    L = emul()
    length = nc if num / den >= 1 else int(round(nc / (num / den)))
    omega, phi = axis(typ, num, den, length, off)
    rng = np.random.default_rng(nc + nout)
    x = rng.random((3, nc), dtype=np.float32) - np.float32(0.5)
    got, P = run(L, x, nout, omega, phi, 1.0 / 7.0)
    assert P >= nc + nout - 1
    ref = series(x, nout, omega, phi, 1.0 / 7.0)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_groups_of_three_interleaved_channels():
    """the x axis of a zoom frame: channel m of row g is a line 3 floats apart in and out"""
    L = emul()
    rows, nc, nout = 9, 200, 700                  # 9 rows: one full run of chan_work's groups of 8 and a tail
    omega, phi = axis(1, 3.5, 1.0, nc, 2.0)
    x = np.random.default_rng(3).random((rows, nc, 3), dtype=np.float32)
    dst = np.full((rows, nout, 3), np.float32(-77))
    p = C.c_void_p()
    assert L.dspfft_cztrows_create(C.byref(p), nc, nout, rows * 3, 3) == 0
    try:
        assert L.dspfft_cztrows_execute(p, x.ctypes.data, nc * 3, 1, 3, dst.ctypes.data, nout * 3, 1, 3, omega, phi, 2.0, None) == 0, L.dspfft_last_error()
        # a second execution with another offset re-uses the chirp's spectrum (same omega)
        omega2, phi2 = axis(1, 3.5, 1.0, nc, -7.5)
        d2 = np.empty_like(dst)
        assert L.dspfft_cztrows_execute(p, x.ctypes.data, nc * 3, 1, 3, d2.ctypes.data, nout * 3, 1, 3, omega2, phi2, 2.0, None) == 0
    finally:
        L.dspfft_cztrows_destroy(p)
    for c in range(3):
        ref = series(x[:, :, c], nout, omega, phi, 2.0)
        assert np.abs(dst[:, :, c] - ref).max() <= 1e-5 * np.abs(ref).max()
        ref2 = series(x[:, :, c], nout, omega2, phi2, 2.0)
        assert np.abs(d2[:, :, c] - ref2).max() <= 1e-5 * np.abs(ref2).max()


def test_transpose_and_refusals():
    L = emul()
    a = np.random.default_rng(1).random((37, 53), dtype=np.float32)
    o = np.full((53, 40), np.float32(-1))
    assert L.dspfft_transpose_f32(o.ctypes.data, 40, a.ctypes.data, 53, 37, 53, None) == 0
    assert np.array_equal(o[:, :37], a.T) and np.all(o[:, 37:] == -1)
    p = C.c_void_p()
    assert L.dspfft_cztrows_create(C.byref(p), 10000, 10000, 1, 1) == -2          # beyond the longest listed convolution
    assert L.dspfft_cztrows_create(C.byref(p), 10, 10, 4, 3) == -1                # lines not a multiple of the group
