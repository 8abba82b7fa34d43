"""spec's encode and ispec's decode pinned to the tools' OWN lines (tests/golden/ref_spec.npz, made by tests/golden/make_ref_fixtures.py from
spec/spec.c:66-139 and spec/ispec.c:66-67,84-87,92-95,98-163 compiled as they lie with COEFF_PRECISION=F, INTERMEDIATE_PRECISION=D): every
range (one / dc / dcs) x scale (log / linear) x sign (abs / shift / saturate / retain) combination, the three gain presets (native, `reference`,
custom) cycled over them, one- and three-channel images, ispec with -p and with a sign map.  tests/test_ref_speclib.py pins the same stages
through include/speclib.c's sibling functions, which hold neither range `dcs`, sign `retain`, the gain presets nor the per-channel maximum.
CPU: the oracle's restatement (oracle/callsite_oracle.c).  -m gpu: dspfft_spec_encode / dspfft_ispec_decode / dspfft_ispec_signmap and the
plan's fused normalisation (spec.c:70-78) through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

FIX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_spec.npz"))
CASES = FIX["cases"]
# reference enum (spec/spec.h:29-47, `none` = 0 first) -> the C ABI's codes (include/dspfft.h): range 0 one 1 dc 2 dcs; scale 0 log 1 linear;
# sign 0 abs 1 shift 2 saturate 3 retain
RANGE = {1: 0, 2: 1, 3: 2}
SCALE = {2: 0, 1: 1}
SIGN = {1: 0, 2: 1, 3: 2, 4: 3}


def case(k):
    h, w, d, r, s, g, gaintype, custom, gain = CASES[k]
    return int(h), int(w), int(d), RANGE[int(r)], SCALE[int(s)], SIGN[int(g)], int(gaintype), float(custom), float(gain)


def close(got, ref, d, sign):
    """float samples of double expressions on both sides: the same float but for a last-place difference where libm's log1p / expm1 and the
    device's differ (<= 2 ulp of the largest sample); the DC pixel of `saturate` is left as it was by both (spec.c:135)"""
    got, ref = got.reshape(-1, d), ref.reshape(-1, d)
    tol = 2.5e-7 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() <= tol, float(np.abs(got.astype(np.float64) - ref).max())


def test_gain_presets_are_the_references():
    """spec.c:81-87: native 127.5 sqrt(4wh), reference 127.5 * 1024, custom as given"""
    seen = set()
    for k in range(len(CASES)):
        h, w, d, _, _, _, gaintype, custom, gain = case(k)
        want = {1: 127.5 * np.sqrt(4.0 * w * h), 2: 127.5 * 1024, 3: custom}[gaintype]
        assert gain == pytest.approx(want, rel=1e-15)
        seen.add(gaintype)
    assert seen == {1, 2, 3}


@pytest.mark.parametrize("k", range(len(CASES)))
def test_normalisation_restatement_is_spec_c_66_78(k):
    h, w, d = case(k)[:3]
    raw = FIX[f"s{k}_raw"].astype(np.float64).reshape(h, w, d)
    f = np.ascontiguousarray(raw.copy())
    ol.lib().oracle_spec_normalise_f64(f.ctypes.data, w, h, d)
    ref = FIX[f"s{k}_normalised"].reshape(h, w, d)
    assert np.abs(f - ref).max() <= 2e-7 * np.abs(ref).max()              # (the reference rounds to float after each of its three divisions)
    assert np.allclose(FIX[f"s{k}_dc"], raw[0, 0] / (4.0 * w * h), rtol=1e-7)     # :66-68 (float DC terms / (w h 4))


@pytest.mark.parametrize("k", range(len(CASES)))
def test_encode_restatement_is_spec_c_81_139(k):
    h, w, d, r, s, g, _, _, gain = case(k)
    O = ol.lib()
    O.oracle_spec_encode_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]
    got = FIX[f"s{k}_normalised"].copy()
    O.oracle_spec_encode_f32(got.ctypes.data, h * w, d, gain, r, s, g)
    close(got, FIX[f"s{k}_encoded"], d, g)


@pytest.mark.parametrize("k", range(len(CASES)))
@pytest.mark.parametrize("pdc", [0, 1])
def test_decode_restatement_is_ispec_c_100_163(k, pdc):
    h, w, d, r, s, g, _, _, gain = case(k)
    O = ol.lib()
    O.oracle_ispec_decode_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    got = FIX[f"s{k}_encoded"].copy()
    DC = FIX[f"s{k}_dc"].copy()
    O.oracle_ispec_decode_f32(got.ctypes.data, h * w, d, gain, r, s, g, DC.ctypes.data, pdc)
    close(unnormalise(got, h, w, d, DC if pdc else None), FIX[f"s{k}_decoded_p{pdc}"], d, g)


def unnormalise(f, h, w, d, dc):
    """ispec.c:153-163 on the decoded samples: first row and column * 1/sqrt2... as the reference does it (P_SQRT2i), / 2, then the DC restore.  The
    product fuses these factors into the inverse plan (dspfft_plan_set_scale / _set_axis_scale0); here they are applied in float like the tool."""
    f = f.reshape(h, w, d).copy()
    r2 = np.float64(np.sqrt(2.0))
    f[0, :, :] = (f[0, :, :].astype(np.float64) * r2).astype(np.float32)
    f[:, 0, :] = (f[:, 0, :].astype(np.float64) * r2).astype(np.float32)
    f = (f.astype(np.float64) / 2).astype(np.float32)
    if dc is not None:
        f[0, 0, :] = dc.astype(np.float32)
    return f.ravel()


def test_p_sqrt2i_is_sqrt2():
    """include/precision.h:127-130: the constant spec.c:71 DIVIDES by and ispec.c:154 MULTIPLIES by is sqrt(2) (M_SQRT2), whatever its name says --
    read off the fixture: normalised first-row samples are raw / sqrt2 / (2 w h)"""
    h, w, d = case(0)[:3]
    raw = FIX["s0_raw"].reshape(h, w, d).astype(np.float64)
    n = FIX["s0_normalised"].reshape(h, w, d)
    assert np.abs(n[0, 3] - raw[0, 3] / np.sqrt(2.0) / (2.0 * w * h)).max() <= 3e-7 * np.abs(n[0, 3]).max()


ABS = [k for k in range(len(CASES)) if f"s{k}_signmap" in FIX.files]


@pytest.mark.parametrize("k", ABS)
def test_signmap_restatement_is_ispec_c_92_95(k):
    """-m: DC[z] = map[z] / 255, every other sample takes the sign of (map - 128); then the decode with -p"""
    h, w, d, r, s, g, _, _, gain = case(k)
    f = FIX[f"s{k}_encoded"].copy()
    m = FIX[f"s{k}_signmap"]
    DC = m[:d].astype(np.float64) / 255.0
    assert np.array_equal(DC, FIX[f"s{k}_dc_signmap"])
    f[d:] = np.copysign(f[d:], m[d:].astype(np.float32) - 128)
    O = ol.lib()
    O.oracle_ispec_decode_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    O.oracle_ispec_decode_f32(f.ctypes.data, h * w, d, gain, r, s, g, DC.ctypes.data, 1)
    close(unnormalise(f, h, w, d, DC), FIX[f"s{k}_decoded_signmap"], d, g)


# ---------------------------------------------------------------- the device kernels through the C ABI
@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dspfun_amd import _lib
    return torch, _lib.load()


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(CASES)))
def test_device_encode_and_decode_are_spec_c_and_ispec_c(gpu, k):
    torch, L = gpu
    h, w, d, r, s, g, _, _, gain = case(k)
    f = torch.from_numpy(FIX[f"s{k}_normalised"].copy()).to("cuda:0")
    assert L.dspfft_spec_encode(f.data_ptr(), h * w, d, gain, r, s, g, None) == 0, L.dspfft_pointwise_last_error()
    torch.cuda.synchronize()
    close(f.cpu().numpy(), FIX[f"s{k}_encoded"], d, g)
    for pdc in (0, 1):
        f = torch.from_numpy(FIX[f"s{k}_encoded"].copy()).to("cuda:0")
        dc = FIX[f"s{k}_dc"].copy()
        assert L.dspfft_ispec_decode(f.data_ptr(), h * w, d, gain, r, s, g, (C.c_double * d)(*dc), pdc, None) == 0, L.dspfft_pointwise_last_error()
        torch.cuda.synchronize()
        close(unnormalise(f.cpu().numpy(), h, w, d, dc if pdc else None), FIX[f"s{k}_decoded_p{pdc}"], d, g)


@pytest.mark.gpu
@pytest.mark.parametrize("k", ABS)
def test_device_signmap_is_ispec_c(gpu, k):
    torch, L = gpu
    h, w, d, r, s, g, _, _, gain = case(k)
    f = torch.from_numpy(FIX[f"s{k}_encoded"].copy()).to("cuda:0")
    m = FIX[f"s{k}_signmap"]
    assert L.dspfft_ispec_signmap(f.data_ptr(), torch.from_numpy(m.copy()).to("cuda:0").data_ptr(), h * w, d, None) == 0
    dc = m[:d].astype(np.float64) / 255.0
    assert L.dspfft_ispec_decode(f.data_ptr(), h * w, d, gain, r, s, g, (C.c_double * d)(*dc), 1, None) == 0
    torch.cuda.synchronize()
    close(unnormalise(f.cpu().numpy(), h, w, d, dc), FIX[f"s{k}_decoded_signmap"], d, g)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 24])
def test_fused_normalisation_of_the_forward_plan_is_spec_c_70_78(gpu, k):
    """what the tool does in three loops after fftw(execute) is the forward plan's scale and per-axis index-0 factors (host/spec_gpu.c); checked on the
    fixture's raw values by running ONLY the scaling: a 1 x 1 ... no transform can be skipped, so the factors are compared on a frame whose transform
    is known -- the unit impulse at (0, 0), whose REDFT10^2 is cos products -- against the reference's normalisation of that same transform"""
    torch, L = gpu
    from dspfun_amd import Plan, REDFT10
    h, w, d = case(k)[:3]
    x = np.zeros((h, w, d), dtype=np.float32); x[0, 0, :] = 1.0
    r2 = np.sqrt(2.0)
    p = Plan.image(h, w, d, REDFT10).set_scale(1.0 / (2.0 * w * h))
    for a in range(2):
        p.set_axis_scale0(a, 1.0, 1.0 / r2)
    t = torch.from_numpy(x).to("cuda:0")
    p.execute(t.data_ptr())
    torch.cuda.synchronize()
    raw = ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10)
    want = np.ascontiguousarray(raw.copy())
    ol.lib().oracle_spec_normalise_f64(want.ctypes.data, w, h, d)          # (held to spec.c:66-78 by test_normalisation_restatement_is_spec_c_66_78)
    assert np.abs(t.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()
