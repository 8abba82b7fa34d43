"""CPU: the device-side scan-method generators (dspfun_amd/csrc/scan_core.h) through the test-only emulation backend, bit-exact
against host/scan_orders.c for every method, including box's out-of-range first leg and cross-index duplicates, ibox's doubled
corner, radial's exact rint(hypot) and magnitude's grouping quirk."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
import scan_device_checks as sd
from emul_lib import emul


def _alloc(n):
    a = np.zeros(max(1, n), dtype=np.uint32)
    return a, a.ctypes.data


def _fetch(a):
    return a.copy()


@pytest.fixture(scope="module")
def so():
    return sd.host_lib()


@pytest.mark.parametrize("w,h", sd.SIZES)
@pytest.mark.parametrize("m", range(len(sd.METHODS)))
def test_generators_match_the_host_library(so, m, w, h):
    sd.check_method(emul(), so, m, w, h, _alloc, _fetch)


def test_box_duplicates_span_frames_on_wide_images(so):
    """16x9: row 8 is re-emitted by every box index >= 8, so those pixels belong to several frames -- the reason box needs stamps"""
    masks = sd.check_method(emul(), so, sd.METHODS.index("box"), 16, 9, _alloc, _fetch)
    total = np.sum(masks, axis=0)
    assert total.max() > 1


@pytest.mark.parametrize("q", [0.0, 40.0])
def test_magnitude_index(q):
    L = emul()
    w, h, ch = 24, 10, 3
    x = ol.synth_f32(5, w * h * ch).reshape(h, w, ch)
    coeffs = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10) / (4.0 * w * h)).astype(np.float32)
    if q:
        coeffs = np.round(coeffs * 8) / 8          # force ties
    idx = np.zeros(w * h, dtype=np.uint32)
    work = np.zeros(L.dspfft_scan_magnitude_work_bytes(w, h), dtype=np.uint8)
    lim = C.c_uint32()
    assert L.dspfft_scan_magnitude_index(idx.ctypes.data, coeffs.ctypes.data, w, h, ch, q, work.ctypes.data, work.size, C.byref(lim), None) == 0
    want, wlim = sd.magnitude_reference(coeffs, w, h, ch, q)
    assert np.array_equal(idx.astype(np.int64), want) and lim.value == wlim
