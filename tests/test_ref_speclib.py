"""The reference's OWN include/speclib.c (compiled where it lies into oracle/_ref/libdspfun_ref.so, COEFF_PRECISION=F /
INTERMEDIATE_PRECISION=D) drives the pins of the spectrogram encode / decode restatements and of the product's device kernels:
  spec_normalization  (speclib.c:79-85)    <-> oracle_spec_normalise_f64 / oracle_scan_normalise_f64  (spec.c:70-78, scan.c:296-298,398)
  spec_scale          (speclib.c:163-170)  <-> oracle_spec_encode_f32  <-> dspfft_spec_encode   (spec.c:81-139; scan.c:374,400)
  spec_unscale        (speclib.c:172-178)  <-> oracle_ispec_decode_f32 <-> dspfft_ispec_decode  (ispec.c:100-151)
for every scale x sign combination speclib covers ("pin what can be pinned").  The float DCT itself stays unpinned
(FFTW is absent); this file pins the elementwise stages either side of it to reference-compiled code."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol


class SpecParams(C.Structure):
    _fields_ = [("scaletype", C.c_int), ("signtype", C.c_int)]       # include/speclib.h:24-27


# (speclib option value, the restatement's / product's code): oracle 0 = log, 1 = linear; 0 = abs, 1 = shift, 2 = saturate
SCALES = [("log", 0), ("linear", 1)]
SIGNS = [("abs", 0), ("shift", 1), ("saturate", 2)]


@pytest.fixture(scope="module")
def ref():
    r = ol.ref()
    if r is None:
        pytest.skip("oracle/_ref not built (reference tree absent and no prebuilt library)")
    r.spec_normalization_pcfi.restype = C.c_double
    r.spec_normalization_pcfi.argtypes = [C.c_size_t]
    r.spec_create_pcfi.restype = C.c_void_p
    r.spec_create_pcfi.argtypes = [C.POINTER(SpecParams), C.c_float, C.c_float]
    r.spec_scale_pcfi.restype = C.c_double
    r.spec_scale_pcfi.argtypes = [C.c_void_p, C.c_double]
    r.spec_unscale.restype = C.c_double
    r.spec_unscale.argtypes = [C.c_void_p, C.c_double]
    r.spec_destroy.argtypes = [C.c_void_p]
    r.spec_param_parse.restype = C.c_char_p
    r.spec_param_parse.argtypes = [C.POINTER(SpecParams), C.c_char_p, C.c_char_p]
    return r


def scaler(ref, scale, sign, mx, gain):
    p = SpecParams(0, 0)
    assert ref.spec_param_parse(C.byref(p), b"scale", scale.encode()) is None
    assert ref.spec_param_parse(C.byref(p), b"sign", sign.encode()) is None
    sp = ref.spec_create_pcfi(C.byref(p), mx, gain)
    assert sp
    return sp


def coefficients(seed, h=9, w=16, d=3):
    """uniform-range DCT coefficients of a synthetic image (what spec.c:78 / scan.c:400 hand to the scaler): |c| <= 1, c[0,0] = mean"""
    x = ol.synth_f32(seed, h * w * d).reshape(h, w, d)
    f = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10))
    ol.lib().oracle_spec_normalise_f64(f.ctypes.data, w, h, d)
    return f.astype(np.float32)


def test_spec_normalization_is_sqrt2_to_the_n_and_matches_both_call_sites(ref):
    for n in range(8):
        assert ref.spec_normalization_pcfi(n) == pytest.approx(np.sqrt(2.0) ** n, rel=1e-15)
    # spec.c:70-78 (first row/column / sqrt2, all / 2wh) == scan.c:296-298,398 (all / 4wh, x spec_normalization_2d(x, y))
    h, w, d = 6, 10, 3
    F = np.ascontiguousarray(ol.dct2d_interleaved(ol.synth_f32(5, h * w * d).reshape(h, w, d).astype(np.float64), ol.REDFT10))
    a, b = F.copy(), F.copy()
    ol.lib().oracle_spec_normalise_f64(a.ctypes.data, w, h, d)
    ol.lib().oracle_scan_normalise_f64(b.ctypes.data, w, h, d)
    for y in range(h):
        for x in range(w):
            b[y, x] *= ref.spec_normalization_pcfi(int(x != 0) + int(y != 0))
    assert np.abs(a - b).max() <= 1e-15 * np.abs(a).max()


@pytest.mark.parametrize("scale,scode", SCALES)
@pytest.mark.parametrize("sign,gcode", SIGNS)
@pytest.mark.parametrize("rangetype", [0, 1])        # "one": max = 1; "dc": max = largest DC over the channels (spec.c:93-100, scan.c:366-374)
def test_reference_spec_scale_pins_the_encode_restatement(ref, scale, scode, sign, gcode, rangetype):
    h, w, d = 9, 16, 3
    c = coefficients(11 + rangetype)
    gain = np.float32(127.5 * np.sqrt(4.0 * w * h))
    mx = np.float32(1.0) if rangetype == 0 else np.float32(c[0, 0].max())
    sp = scaler(ref, scale, sign, mx, gain)
    want = np.array([ref.spec_scale_pcfi(sp, float(v)) for v in c.ravel()]).reshape(c.shape)
    got = c.copy()
    O = ol.lib()
    O.oracle_spec_encode_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]
    O.oracle_spec_encode_f32(got.ctypes.data, h * w, d, float(gain), rangetype, scode, gcode)
    sl = slice(1, None) if sign == "saturate" else slice(None)     # spec.c:133 leaves the DC pixel out of saturate
    assert np.abs(got.reshape(-1, d)[sl] - want.reshape(-1, d)[sl]).max() <= 3e-6
    # and back: spec_unscale vs the decode restatement on the reference's own encoding
    enc = want.astype(np.float32)
    back_ref = np.array([ref.spec_unscale(sp, float(v)) for v in enc.ravel()]).reshape(c.shape)
    dec = enc.copy()
    DC = c[0, 0].astype(np.float64).copy()
    O.oracle_ispec_decode_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    O.oracle_ispec_decode_f32(dec.ctypes.data, h * w, d, float(gain), rangetype, scode, gcode, DC.ctypes.data, 0)
    assert np.abs(dec.reshape(-1, d)[sl] - back_ref.reshape(-1, d)[sl]).max() <= 3e-6 * max(1.0, np.abs(back_ref).max())
    if sign == "shift":                                  # the only sign mode that keeps the information: decode(encode(c)) == c
        assert np.abs(back_ref - c).max() <= 1e-6
    ref.spec_destroy(sp)


@pytest.mark.gpu
@pytest.mark.parametrize("scale,scode", SCALES)
@pytest.mark.parametrize("sign,gcode", SIGNS)
def test_reference_spec_scale_pins_the_device_kernels(ref, scale, scode, sign, gcode):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dspfun_amd import _lib
    L = _lib.load()
    h, w, d = 45, 64, 3
    c = coefficients(23, h, w, d)
    gain = np.float32(127.5 * np.sqrt(4.0 * w * h))
    for rangetype in (0, 1):
        mx = np.float32(1.0) if rangetype == 0 else np.float32(c[0, 0].max())
        sp = scaler(ref, scale, sign, mx, gain)
        want = np.array([ref.spec_scale_pcfi(sp, float(v)) for v in c.ravel()]).reshape(c.shape)
        f = torch.from_numpy(c.copy()).to("cuda:0")
        assert L.dspfft_spec_encode(f.data_ptr(), h * w, d, float(gain), rangetype, scode, gcode, None) == 0
        torch.cuda.synchronize()
        sl = slice(1, None) if sign == "saturate" else slice(None)
        assert np.abs(f.cpu().numpy().reshape(-1, d)[sl] - want.reshape(-1, d)[sl]).max() <= 3e-6
        back_ref = np.array([ref.spec_unscale(sp, float(v)) for v in f.cpu().numpy().ravel()]).reshape(c.shape)
        DC = (C.c_double * d)(*c[0, 0].astype(np.float64))
        assert L.dspfft_ispec_decode(f.data_ptr(), h * w, d, float(gain), rangetype, scode, gcode, DC, 0, None) == 0
        torch.cuda.synchronize()
        assert np.abs(f.cpu().numpy().reshape(-1, d)[sl] - back_ref.reshape(-1, d)[sl]).max() <= 3e-6 * max(1.0, np.abs(back_ref).max())
        ref.spec_destroy(sp)
