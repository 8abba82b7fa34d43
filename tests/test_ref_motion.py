"""CPU: motion's elementwise stages pinned to the reference's own compiled lines (tests/golden/ref_motion.npz, made by
tests/golden/make_ref_fixtures.py from motion/motion.c:559-573,644-647,650,683-744,748-751,756-776 with COEFF_PRECISION=F,
INTERMEDIATE_PRECISION=L as motion/Makefile:1-2 builds): uniform-range scaling, the six-face damp / boost, threshold, DC preservation,
the quantiser with its count of coded coefficients, the reverse scaling, the 8-bit store.  Checked here: the oracle's restatement
(oracle/callsite_oracle.c, tests/motion_ref.py) and the per-element functions the HIP kernels call (motion_filter.h, elementwise_core.h)
through the test-only emulation library.  The GPU kernels themselves: tests/test_ref_motion_gpu.py."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from emul_lib import emul

FIX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_motion.npz"))
NCASES = len(FIX["cases"])
I3, I2 = C.c_int * 3, C.c_int * 2


def case(ci):
    """the arguments of dspfft_motion_filter (include/dspfft.h) for fixture case ci, derived as motion.c:566-572,736 derives them"""
    r = FIX["cases"][ci]
    d, h, w, md, mh, mw = (int(v) for v in r[:6])
    b0, b1 = [int(v) for v in r[6:9]], [int(v) for v in r[9:12]]
    damp, boost, tmin, tmax, pdc, quant = float(r[12]), float(r[13]), float(r[14]), float(r[15]), int(r[16]), float(r[17])
    whd8 = np.longdouble(8) * w * h * d                      # 1 / normalization^2
    dcstop = any(b0)
    return dict(active=(d, h, w), minbuf=(md, mh, mw), band_begin=b0, band_end=b1, damp=damp, boost=boost,
                threshold_lo=float(np.float32(np.longdouble(tmin) * 255 * whd8)), threshold_hi=float(np.float32(np.longdouble(tmax) * 255 * whd8)),
                preserve_dc=pdc, grey_add=float(np.float32((1 - np.longdouble(damp if dcstop else boost)) * np.longdouble(127.5) * whd8)),
                quantizer=float(np.float32(np.longdouble(quant) * 8 * np.sqrt(np.longdouble(w * h * d)))), quant=quant)


def run_filter(fn, ci, restype=C.c_ulonglong):
    p = case(ci)
    buf = FIX[f"m{ci}_in"].copy()
    fn.restype = restype
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                   C.c_double if "oracle" in fn.__name__ else C.c_float, C.c_float]
    coded = fn(buf.ctypes.data, I3(*p["active"]), I2(p["minbuf"][1], p["minbuf"][2]), I3(*p["band_begin"]), I3(*p["band_end"]), p["damp"], p["boost"],
               p["threshold_lo"], p["threshold_hi"], p["preserve_dc"], p["grey_add"], p["quantizer"])
    return buf, int(coded), p


def check_filtered(got, coded, p, ci):
    ref = FIX[f"m{ci}_filtered"]
    if p["damp"] == 1.0 and p["boost"] == 1.0 and p["preserve_dc"] != 2:
        # nothing but exact steps (threshold, dc copy, the quantiser): bit for bit, and the same count of coded coefficients
        assert np.array_equal(got, ref), int((got != ref).sum())
    else:
        # damp / boost / grey are `intermediate` (long double) factors in the reference and floats in the C ABI: one rounding apart
        assert np.abs(got.astype(np.float64) - ref).max() <= 2e-7 * np.abs(ref).max()
        if p["quant"]:      # behind the quantiser a last-bit difference either vanishes or moves a value by one step, rarely
            assert (got != ref).sum() <= 2
    if p["quant"]:
        assert abs(coded - int(FIX[f"m{ci}_filtered_coded"][0])) <= (0 if p["damp"] == 1.0 and p["boost"] == 1.0 else 2)


@pytest.mark.parametrize("ci", range(NCASES))
def test_oracle_filter_is_the_references(ci):
    got, coded, p = run_filter(ol.lib().oracle_motion_filter_f32, ci)
    check_filtered(got, coded, p, ci)


@pytest.mark.parametrize("ci", range(NCASES))
def test_kernel_filter_function_is_the_references(ci):
    """motion_filter.h motion_filter_at -- what pointwise.hip's kernel and the fused column roundtrip call per element"""
    L = C.CDLL(emul()._name)
    got, coded, p = run_filter(L.emul_motion_filter, ci)
    check_filtered(got, coded, p, ci)


@pytest.mark.parametrize("ci", range(NCASES))
def test_uniform_range_scaling_is_the_plans_scale_and_axis_scale0(ci):
    """motion.c:644-647 / :748-751: what the forward plan's set_scale(2 sqrt2 ...) + set_axis_scale0(a, 1, 1/sqrt2) and the inverse plan's
    mirror image multiply by (host/motion_gpu.c, tools/bench_motion.py): 2 sqrt2 / prod(sqrt2 where the index is 0) over the ACTIVE block"""
    p = case(ci)
    d, h, w = p["active"]
    md, mh, mw = p["minbuf"]
    x = FIX[f"m{ci}_in"].reshape(md, mh, mw).astype(np.float64)
    f = np.full((d, h, w), 2 * np.sqrt(2.0))
    f[0, :, :] /= np.sqrt(2.0); f[:, 0, :] /= np.sqrt(2.0); f[:, :, 0] /= np.sqrt(2.0)
    want = x.copy(); want[:d, :h, :w] *= f
    got = FIX[f"m{ci}_scaled"].reshape(md, mh, mw)
    assert np.abs(got - want).max() <= 1.2e-7 * np.abs(want).max()
    assert np.array_equal(got[d:], x[d:].astype(np.float32)) and np.array_equal(got[:, h:], x[:, h:].astype(np.float32))      # the embedding is untouched
    back = x.copy(); back[:d, :h, :w] /= f
    assert np.abs(FIX[f"m{ci}_unscaled"].reshape(md, mh, mw) - back).max() <= 1.2e-7 * np.abs(back).max()


@pytest.mark.parametrize("ci", range(NCASES))
def test_8bit_store_is_the_references(ci):
    """motion.c:759,766,776: pel = coeff * scalefactor * normalization^2, clamped and lround()ed -- dspfft_f32_to_u8(dst, src, mul) with
    mul = scalefactor * normalization^2 (elementwise_core.h quantise_u8), over the active block"""
    p = case(ci)
    d, h, w = p["active"]
    md, mh, mw = p["minbuf"]
    src = FIX[f"m{ci}_store_in"]
    L = emul()
    dst = np.zeros(src.size, dtype=np.uint8)
    assert L.dspfft_f32_to_u8(dst.ctypes.data, src.ctypes.data, 1.0 / (8.0 * w * h * d), src.size, None) == 0
    ref = FIX[f"m{ci}_store_u8"].reshape(md, mh, mw)
    got = dst.reshape(md, mh, mw)
    # the first 16 inputs are exact halves (0.5 ... 15.5) BEFORE the reference multiplies by its long double normalization twice (1/sqrtl is
    # inexact: 8.5 arrives as 8.4999999999999999 and goes down); the kernels multiply once by scalefactor * normalization^2.  At such ties the two
    # may differ by one level ("exact except <= 1 LSB ties"); everywhere else they are equal.
    g, r = got[:d, :h, :w].ravel(), ref[:d, :h, :w].ravel()
    assert np.array_equal(g[16:], r[16:])
    assert np.abs(g[:16].astype(int) - r[:16].astype(int)).max() <= 1
    assert ref[:d, :h, :w].min() == 0 and ref[:d, :h, :w].max() == 255          # the clamps were exercised


@pytest.mark.parametrize("q", [1440.0, 3.0, 20.0 * 8 * float(np.sqrt(1920 * 1080.0)), float(np.nextafter(np.float32(2), np.float32(0))), float(np.nextafter(np.float32(1), np.float32(2))),
                               0.3, 7e-4, 1.0, 65537.0, 1e-35])
def test_quantiser_without_a_divider_is_the_float_division(q):
    """motion_filter.h motion_quantise: two residual steps on v * RN(1/q) reproduce the correctly rounded FLOAT quotient of motion.c:744, so
    round(v / q) * q comes out bit for bit -- over 2^22 values per quantiser across sixty binades, half-integers' neighbourhoods included; the
    last quantiser lies outside the range the reciprocal is used for and takes the division itself"""
    L = C.CDLL(emul()._name)
    q32 = np.float32(q)
    n = 1 << 22
    rng = np.random.default_rng(int(q32.view(np.uint32)))
    v = (rng.standard_normal(n) * np.exp2(rng.integers(-30, 30, n))).astype(np.float32) * q32
    k = rng.integers(-4096, 4096, n // 4).astype(np.float32) + np.float32(0.5)               # quotients at and next to the halves
    v[: n // 4] = k * q32
    v[n // 4: n // 2] = np.nextafter(k * q32, np.float32(np.inf))
    v[n // 2: 3 * n // 4] = np.nextafter(k * q32, np.float32(-np.inf))
    v = v[np.isfinite(v)]
    t = (v / q32).astype(np.float32)                                                          # IEEE float division
    r = np.trunc(t.astype(np.float64) + np.copysign(0.5, t)).astype(np.float32)              # roundf: halves away from zero
    want = (r * q32).astype(np.float32)
    ok = np.isfinite(want)
    v, want = v[ok], want[ok]
    got = v.copy()
    L.emul_motion_filter.restype = C.c_ulonglong
    L.emul_motion_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
    coded = L.emul_motion_filter(got.ctypes.data, I3(1, 1, got.size), I2(1, got.size), I3(0, 0, 0), I3(1, 1, got.size), 1.0, 1.0, 0.0, 0.0, 0, 0.0, float(q32))
    assert np.array_equal(got, want), int((got != want).sum())
    assert coded == int((want != 0).sum())


@pytest.mark.parametrize("mul", [1.0 / (8.0 * 1920 * 1080), 1.0 / 3840.0, 1.0, 0.3337, 255.0, 1.0 / (8.0 * 1920 * 1080 * 256)])
def test_fused_8bit_store_is_the_double_path(mul):
    """elementwise_core.h quantise_u8_of (what the rows' 8-bit end and the block kernels store): v * (float)mul rounded in single precision unless it
    lies next to a rounding boundary, where motion.c:776's double expression decides -- the same byte as quantise_u8((double)v * mul) for every
    value: 2^22 of them per multiplier, with pixel values at, just below and just above every half-integer of 0 ... 256, the clamps' neighbourhoods,
    huge values, infinities and NaN"""
    L = C.CDLL(emul()._name)
    n = 1 << 22
    rng = np.random.default_rng(int(np.float64(mul).view(np.uint64) % (1 << 32)))
    pel = rng.uniform(-8.0, 264.0, n)
    halves = rng.integers(-4, 520, n // 2) * 0.5                                             # k / 2: integers and halves
    pel[: n // 2] = halves
    v = (pel / mul).astype(np.float32)
    v[: n // 6] = np.nextafter(v[: n // 6], np.float32(np.inf))
    v[n // 6: n // 3] = np.nextafter(v[n // 6: n // 3], np.float32(-np.inf))
    v[-8:] = np.array([np.inf, -np.inf, np.nan, 3e38, -3e38, 0.0, -0.0, 1e-45], dtype=np.float32)
    fast = np.zeros(n, dtype=np.uint8); exact = np.zeros(n, dtype=np.uint8)
    L.emul_quantise_u8_of.restype = None
    L.emul_quantise_u8_of.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_longlong]
    L.emul_quantise_u8_of(v.ctypes.data, mul, fast.ctypes.data, exact.ctypes.data, n)
    ok = ~np.isnan(v)                                                                       # (NaN: the double path's conversion is the platform's; the device gives 0)
    assert np.array_equal(fast[ok], exact[ok]), int((fast[ok] != exact[ok]).sum())
    assert fast[np.isnan(v)].max() == 0
    assert exact.min() == 0 and exact.max() == 255 and len(np.unique(exact)) == 256


# ---- the pixel load (motion.c:617-638) and the output stage (:755-776) with every --ispec / --spec mode, 8-bit and float pixels ----
IO_GEOMS = [tuple(int(v) for v in r) for r in FIX["io_geoms"]]
ISPEC_MODES = {"none": 0, "shift": 2, "flat": 3, "copy": 4}                 # DSPFFT_MOTION_* (include/dspfft.h)
SPEC_MODES = {"none": 0, "abs": 1, "shift": 2, "flat": 3, "copy": 4}


def io_consts(g):
    """scalefactor, normalization (motion.c:566-567; block == scaled in the fixtures) and the constants :568-569,755 as the reference computed them"""
    d, h, w = IO_GEOMS[g][:3]
    return 1.0, float(1 / np.sqrt(np.longdouble(w * h * d * 8))), {"shift": float(FIX[f"io{g}_c_shift"][0]), "abs": float(FIX[f"io{g}_c_abs"][0])}, float(FIX[f"io{g}_ic"][0])


def corner(a, g):
    d, h, w, md, mh, mw = IO_GEOMS[g]
    return a.reshape(md, mh, mw)[:d, :h, :w]


def check_load(got, g, name, px):
    """coefficients are `coeff` (float) values of long double expressions in the reference, of double ones here: the same float but for the rare
    value whose two roundings straddle; outside the block the reference's buffer is zero (:617) and the load writes nothing there"""
    ref = corner(FIX[f"io{g}_load_{name}_{px}"], g).astype(np.float64)
    err = np.abs(corner(got, g).astype(np.float64) - ref)
    assert (err <= np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)).all(), (g, name, px, err.max())
    assert (err > 0).mean() <= 0.01


def check_store(got, g, name, px):
    ref = corner(FIX[f"io{g}_store_{name}_{px}"], g)
    out = corner(got, g)
    if px == "u8":
        diff = np.abs(out.astype(int) - ref.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() <= 0.002, (g, name, diff.max(), (diff > 0).mean())      # (a pel within 1e-16 of a half rounds either way)
    else:
        err = np.abs(out.astype(np.float64) - ref.astype(np.float64))
        assert (err <= np.spacing(np.abs(ref))).all() and (err > 0).mean() <= 0.01, (g, name, err.max())


@pytest.mark.parametrize("g", range(len(IO_GEOMS)))
@pytest.mark.parametrize("px", ["u8", "f32"])
@pytest.mark.parametrize("name", list(ISPEC_MODES))
def test_load_functions_of_the_kernels_are_the_references(g, px, name):
    d, h, w, md, mh, mw = IO_GEOMS[g]
    _, norm, _, ic = io_consts(g)
    E = emul()
    E.emul_motion_load.restype = None
    E.emul_motion_load.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double]
    pix = FIX[f"io{g}_pix_{px}"].copy()
    c = np.zeros(md * mh * mw, dtype=np.float32)
    E.emul_motion_load(c.ctypes.data, pix.ctypes.data, int(px == "f32"), I3(d, h, w), I2(mh, mw), ISPEC_MODES[name], ic, norm)
    check_load(c, g, name, px)
    assert not corner(c, g)[..., :0].size and c.reshape(md, mh, mw)[:, h:, :].max(initial=0) == 0 and c.reshape(md, mh, mw)[:, :, w:].max(initial=0) == 0


@pytest.mark.parametrize("g", range(len(IO_GEOMS)))
@pytest.mark.parametrize("px", ["u8", "f32"])
@pytest.mark.parametrize("name", list(SPEC_MODES))
def test_store_functions_of_the_kernels_are_the_references(g, px, name):
    d, h, w, md, mh, mw = IO_GEOMS[g]
    sf, norm, cc, _ = io_consts(g)
    E = emul()
    E.emul_motion_store.restype = None
    E.emul_motion_store.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
    out = np.zeros(md * mh * mw, dtype=np.float32 if px == "f32" else np.uint8)
    coeffs = FIX[f"io{g}_coeffs"]                 # (kept alive across the call: the archive hands out a new array per access)
    E.emul_motion_store(out.ctypes.data, int(px == "f32"), coeffs.ctypes.data, I3(d, h, w), I2(mh, mw), SPEC_MODES[name], sf, norm, cc.get(name, 0.0))
    check_store(out, g, name, px)


def test_spectrogram_constants_are_the_references():
    """c of --spec shift / ic of --ispec shift (motion.c:568-569) and c of --spec abs from the block's DC (:755), as tests/motion_ref.py restates them"""
    for g in range(len(IO_GEOMS)):
        d, h, w = IO_GEOMS[g][:3]
        sf, norm, cc, ic = io_consts(g)
        want = 127.5 / np.log1p(w * h * d * norm * 255 * 8)
        assert abs(cc["shift"] - want) <= 1e-12 * want and abs(ic - want) <= 1e-12 * want
        dc = float(FIX[f"io{g}_coeffs"][0])
        assert abs(cc["abs"] - 255.0 / np.log1p(abs(dc * sf * norm))) <= 1e-12 * cc["abs"]
