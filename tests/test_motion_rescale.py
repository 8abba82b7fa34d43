"""motion with scaled != block, top-N and the spectrogram outputs: the device pipeline
(dspfft_execute_roundtrip_u8 over two plans of different extents in one embedding, dspfft_motion_topn, dspfft_motion_load_u8 /
_store_u8) against the f64 restatement tests/motion_ref.py.  The rescale path of the engine also runs on the CPU emulation."""
import ctypes as C
import math

import numpy as np
import pytest

import motion_ref as mr
import oracle_lib as ol


def plans(Plan, block, scaled, minbuf, lib=None):
    from dspfun_amd import REDFT10, REDFT01
    r2 = math.sqrt(2.0)
    fwd = Plan.many_r2r(list(block), [REDFT10] * 3, inembed=list(minbuf), onembed=list(minbuf), lib=lib).set_scale(2 * r2)
    inv = Plan.many_r2r(list(scaled), [REDFT01] * 3, inembed=list(minbuf), onembed=list(minbuf), lib=lib).set_scale(1.0 / (2 * r2))
    for a in range(3):
        fwd.set_axis_scale0(a, 1.0, 1.0 / r2)
        inv.set_axis_scale0(a, r2, 1.0)
    return fwd, inv


CASES = [((4, 12, 16), (4, 18, 24)),        # upscale in y and x: zero-padded spectrum
         ((6, 20, 24), (3, 10, 16)),        # downscale on every axis: truncated spectrum
         ((4, 10, 12), (6, 8, 18))]         # mixed


@pytest.mark.parametrize("block,scaled", CASES)
def test_rescale_roundtrip_on_the_emulation(block, scaled):
    from dspfun_amd.engine import Plan
    from emul_lib import emul
    minbuf = tuple(max(b, s) for b, s in zip(block, scaled))
    pix = ol.synth_u8(41, int(np.prod(minbuf))).reshape(minbuf)
    fwd, inv = plans(Plan, block, scaled, minbuf, lib=emul())
    scalefactor, normalization = mr.consts(block, scaled)
    work = np.full(minbuf, 7.0, dtype=np.float32)                # must be zeroed by the call
    out = np.zeros(minbuf, dtype=np.uint8)
    fwd.roundtrip_u8(inv, pix.ctypes.data, out.ctypes.data, work.ctypes.data, scalefactor * normalization * normalization)
    want, _, _ = mr.block_roundtrip(pix, block, scaled, minbuf, impl="direct")
    sd, sh, sw = scaled
    diff = np.abs(out[:sd, :sh, :sw].astype(int) - want[:sd, :sh, :sw].astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01          # <= 1 LSB where the f64 value sits on a rounding boundary
    assert not out[sd:].any() and not out[:, sh:].any() and not out[:, :, sw:].any()      # nothing outside the scaled region is written


@pytest.mark.gpu
@pytest.mark.parametrize("block,scaled", CASES + [((8, 54, 96), (8, 108, 192)), ((16, 108, 192), (8, 54, 96))])
@pytest.mark.parametrize("quant", [0.0, 0.4])
def test_rescale_roundtrip_gpu(block, scaled, quant):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dspfun_amd import Plan
    minbuf = tuple(max(b, s) for b, s in zip(block, scaled))
    active = tuple(min(b, s) for b, s in zip(block, scaled))
    pix = ol.synth_u8(43, int(np.prod(minbuf))).reshape(minbuf)
    fwd, inv = plans(Plan, block, scaled, minbuf)
    scalefactor, normalization = mr.consts(block, scaled)
    d_pix = torch.from_numpy(pix).to("cuda:0")
    d_out = torch.zeros_like(d_pix)
    work = torch.full(minbuf, 3.0, dtype=torch.float32, device="cuda:0")
    flt = None
    if quant:
        flt = dict(active=active, minbuf_hw=minbuf[1:], block_depth=minbuf[0], band_begin=(0, 0, 0), band_end=active,
                   quantizer=quant * 8 * math.sqrt(float(np.prod(scaled))))
    fwd.roundtrip_u8(inv, d_pix.data_ptr(), d_out.data_ptr(), work.data_ptr(), scalefactor * normalization * normalization, filter=flt)
    torch.cuda.synchronize()
    want, _, _ = mr.block_roundtrip(pix, block, scaled, minbuf, quant=quant)
    sd, sh, sw = scaled
    got = d_out.cpu().numpy()
    diff = np.abs(got[:sd, :sh, :sw].astype(int) - want[:sd, :sh, :sw].astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < (0.02 if quant else 0.005), (diff.max(), (diff > 0).mean())
    if not quant and all(s >= b for s, b in zip(scaled, block)) and all(s % b == 0 for s, b in zip(scaled, block)):
        pass    # (an integer upscale interpolates; the original samples are not reproduced at integer positions for DCT-II grids)


@pytest.mark.gpu
@pytest.mark.parametrize("keep", [1, 37, 5000])
def test_topn_select_gpu(keep):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from dspfun_amd import _lib
    L = _lib.load()
    n = 200_000
    c = (ol.synth_f32(99, n) - 0.5).astype(np.float32)
    c[::7] = np.round(c[::7] * 16) / 16                 # plenty of exact ties, also at the threshold for some `keep`
    if keep == 37:
        c[np.argsort(-np.abs(c))[30:45]] = 0.4375       # force the threshold to fall inside a run of equal magnitudes
    d = torch.from_numpy(c.copy()).to("cuda:0")
    work = torch.zeros(L.dspfft_motion_topn_work_bytes(n), dtype=torch.uint8, device="cuda:0")
    assert L.dspfft_motion_topn(d.data_ptr(), n, keep, work.data_ptr(), work.numel(), None) == 0, L.dspfft_motion_last_error()
    torch.cuda.synchronize()
    order = np.argsort(-np.abs(c), kind="stable")
    want = np.zeros_like(c)
    want[order[:keep]] = c[order[:keep]]
    assert np.array_equal(d.cpu().numpy(), want)
    assert int((d != 0).sum()) <= keep


@pytest.mark.gpu
@pytest.mark.parametrize("spec", ["abs", "shift", "flat", "none"])
def test_motion_spectrogram_store_and_ispec_load_gpu(spec):
    """motion --spec: forward, uniform range, then the spectrogram encode of :755-776 instead of the inverse; and --ispec decodes the
    8-bit spectrogram back (:626-630) up to its quantisation"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from dspfun_amd import Plan, _lib
    L = _lib.load()
    block = scaled = minbuf = (4, 30, 40)
    pix = ol.synth_u8(45, int(np.prod(minbuf))).reshape(minbuf)
    fwd, inv = plans(Plan, block, scaled, minbuf)
    scalefactor, normalization = mr.consts(block, scaled)
    I3, I2 = (C.c_int * 3)(*block), (C.c_int * 2)(*minbuf[1:])
    d_pix = torch.from_numpy(pix).to("cuda:0")
    cbuf = torch.zeros(minbuf, dtype=torch.float32, device="cuda:0")
    assert L.dspfft_motion_load_u8(cbuf.data_ptr(), d_pix.data_ptr(), I3, I2, 0, 0.0, normalization, None) == 0
    fwd.execute(cbuf.data_ptr())
    torch.cuda.synchronize()
    want, coeffs, cc = mr.block_roundtrip(pix, block, scaled, minbuf, spec=spec)
    assert np.abs(cbuf.cpu().numpy() - coeffs).max() <= 1e-5 * np.abs(coeffs).max()
    mode = {"none": 0, "abs": 1, "shift": 2, "flat": 3}[spec]
    if spec == "none":
        inv.execute(cbuf.data_ptr())
    out = torch.zeros_like(d_pix)
    assert L.dspfft_motion_store_u8(out.data_ptr(), cbuf.data_ptr(), I3, I2, mode, scalefactor, normalization, cc, None) == 0
    torch.cuda.synchronize()
    diff = np.abs(out.cpu().numpy().astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01
    if spec in ("shift", "flat"):           # decode the spectrogram again: :626-630 invert :763-764 up to the 8-bit rounding
        back = torch.zeros(minbuf, dtype=torch.float32, device="cuda:0")
        assert L.dspfft_motion_load_u8(back.data_ptr(), out.data_ptr(), I3, I2, mode, cc, normalization, None) == 0
        torch.cuda.synchronize()
        b = back.cpu().numpy().astype(np.float64)
        o = out.cpu().numpy().astype(np.float64)
        if spec == "flat":
            exp = (o - 127.5) * 2 / normalization / normalization
        else:
            exp = np.copysign(np.expm1(np.abs((o - 127.5) / cc)), o - 127.5) / normalization
        assert np.abs(b - exp).max() <= 1e-5 * max(1.0, np.abs(exp).max())
