"""GPU (-m gpu): BASELINE configs 4 and 5 at their FULL sizes ("configs exercised only in reduced form").

C4  scan zigzag progressive reconstruction of 7680x4320 RGB, step 2^20 -> 32 output frames (scan/scan.c:292-298,347-350,377-383,
    421-459): forward with 1/(4wh) fused, zigzag frame ids on the device, 32 fused masked-accumulate steps; the first two frames
    against the f64 CPU port of the same step, the final sum against the input (<= 5e-6).
C5  motion's 3-D blocks at the clip's plane sizes (motion/motion.c:61-67,535-552,617-647,748-776): chroma 960x540x256 and luma
    1920x1080x256; u8 roundtrip exact, forward 3-D coefficients against the f64 port at full size (memory permitting)."""
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dspfun_amd import _lib
    _lib.load()
    return torch


def host_threads():
    return max(1, min(ol.lib().cpu_port_max_threads(), os.cpu_count() or 1, 64))


def free_host_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 2 ** 30
    except Exception:
        return 0.0


def test_c4_full_workload_8k_step_2pow20(gpu):
    from dspfun_amd import Plan, _lib, REDFT10, REDFT01
    L = _lib.load()
    w, h, c = 7680, 4320, 3
    step = 1 << 20
    nframes = (w * h + step - 1) // step
    assert nframes == 32
    x = ol.synth_f32(0xD5F0004, w * h * c).reshape(h, w, c)
    coeffs = gpu.from_numpy(x).to("cuda:0")
    Plan.image(h, w, c, REDFT10).set_scale(1.0 / (4.0 * w * h)).execute(coeffs.data_ptr())
    inv = Plan.image(h, w, c, REDFT01)
    ids = gpu.zeros(w * h, dtype=gpu.int32, device="cuda:0")
    assert L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, step, None) == 0
    acc = gpu.empty_like(coeffs)
    work = gpu.empty_like(coeffs)
    assert L.dspfft_broadcast_dc(acc.data_ptr(), coeffs.data_ptr(), w * h, c, None) == 0
    # the device's owner ids are the zigzag order cut into steps (integer, bit-exact): id of pixel zz[i] = i // step, DC excluded
    zz = ol.zigzag_order(w, h).astype(np.int64)
    frame_of = np.empty(w * h, dtype=np.int32)
    frame_of[zz] = (np.arange(w * h, dtype=np.int64) // step).astype(np.int32)
    frame_of[0] = -1
    got_ids = ids.cpu().numpy()
    assert np.array_equal(got_ids[1:], frame_of[1:]) and got_ids[0] == -1
    del zz, got_ids
    thr = host_threads()
    cf64 = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=thr))
    ol.lib().oracle_scan_normalise_f64(cf64.ctypes.data, w, h, c)
    # the forward coefficients themselves (config 4's DCT-II at full size)
    got = coeffs.cpu().numpy()
    assert np.abs(got - cf64).max() <= 1e-5 * np.abs(cf64).max()
    del got
    ref = np.ascontiguousarray(np.broadcast_to(cf64[0, 0], (h, w, c)).copy())
    inv.scan_prepare(ids.data_ptr(), c)
    skipped = []
    for f in range(nframes):
        work.fill_(float("nan"))       # tiles the masked column pass skips (no coefficient of this frame) must not be read back
        inv.execute_masked_accumulate(coeffs.data_ptr(), work.data_ptr(), acc.data_ptr(), ids.data_ptr(), f, c)
        skipped.append(float(gpu.isnan(work).float().mean()))
        if f < 2:
            rec = np.where((frame_of.reshape(h, w) == f)[:, :, None], cf64, 0.0)
            ref += ol.dct2d_interleaved(rec, REDFT01, impl="port", threads=thr)
            del rec
            gpu.cuda.synchronize()
            assert np.abs(acc.cpu().numpy() - ref).max() < 5e-6, f
    gpu.cuda.synchronize()
    assert float((acc.cpu() - gpu.from_numpy(x)).abs().max()) <= 5e-6
    # zigzag frames of a 16:9 image: the first touches 19 % of the columns, the middle ones 59 %
    assert skipped[0] > 0.7 and 0.3 < float(np.mean(skipped)) < 0.7, skipped


@pytest.mark.parametrize("w,h", [(3840, 2160), (1920, 1080)])
def test_sparse_frames_skip_matches_dense(gpu, w, h, monkeypatch):
    """the fused scan step with and without the empty-tile skip (DSPFFT_NO_ZSKIP=1), zigzag, 9 frames"""
    from dspfun_amd import Plan, _lib, REDFT10, REDFT01
    L = _lib.load()
    monkeypatch.setenv("DSPFFT_ZSKIP", "1")           # plain plans (no column split at these sizes) take part on request only
    c, nframes = 3, 9
    x = ol.synth_f32(w + h, w * h * c).reshape(h, w, c)
    coeffs = gpu.from_numpy(x).to("cuda:0")
    Plan.image(h, w, c, REDFT10).set_scale(1.0 / (4.0 * w * h)).execute(coeffs.data_ptr())
    ids = gpu.zeros(w * h, dtype=gpu.int32, device="cuda:0")
    assert L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, (w * h + nframes - 1) // nframes, None) == 0
    inv = Plan.image(h, w, c, REDFT01)
    out = {}
    for skip in (True, False, "prepared"):
        if skip:
            monkeypatch.delenv("DSPFFT_NO_ZSKIP", raising=False)
        else:
            monkeypatch.setenv("DSPFFT_NO_ZSKIP", "1")
        inv.scan_prepare(ids.data_ptr() if skip == "prepared" else 0, c)
        acc = gpu.empty_like(coeffs); work = gpu.empty_like(coeffs)
        assert L.dspfft_broadcast_dc(acc.data_ptr(), coeffs.data_ptr(), w * h, c, None) == 0
        sums = []
        for f in range(nframes):
            work.fill_(float("nan"))
            inv.execute_masked_accumulate(coeffs.data_ptr(), work.data_ptr(), acc.data_ptr(), ids.data_ptr(), f, c)
            sums.append(acc.clone())
        out[skip] = sums
        gpu.cuda.synchronize()
        assert (float(gpu.isnan(work).float().mean()) > 0.2) == bool(skip)     # the last frame: the far corner of the spectrum
    for a, b, p in zip(out[True], out[False], out["prepared"]):
        assert float((a - b).abs().max()) < 2e-6
        assert gpu.equal(a, p)
    assert float((out[True][-1].cpu() - gpu.from_numpy(x)).abs().max()) <= 5e-6


def _plans3d(d, h, w):
    from dspfun_amd import Plan, REDFT10, REDFT01
    r2 = float(np.sqrt(2.0))
    fwd = Plan.many_r2r([d, h, w], [REDFT10] * 3).set_scale(2 * r2)
    inv = Plan.many_r2r([d, h, w], [REDFT01] * 3).set_scale(1.0 / (2 * r2))
    for a in range(3):
        fwd.set_axis_scale0(a, 1.0, 1.0 / r2)
        inv.set_axis_scale0(a, r2, 1.0)
    return fwd, inv


@pytest.mark.parametrize("plane,h,w", [("chroma", 540, 960), ("luma", 1080, 1920)])
def test_c5_plane_full_size_3d_coefficients_and_u8_roundtrip(gpu, plane, h, w):
    from dspfun_amd import _lib
    L = _lib.load()
    d = 256
    n = d * h * w
    pix_h = ol.synth_u8(0xD5F0005 + (plane == "chroma"), n).reshape(d, h, w)
    pix = gpu.from_numpy(pix_h).to("cuda:0")
    cbuf = gpu.empty((d, h, w), dtype=gpu.float32, device="cuda:0")
    assert L.dspfft_u8_to_f32(cbuf.data_ptr(), pix.data_ptr(), n, None) == 0
    fwd, inv = _plans3d(d, h, w)
    fwd.execute(cbuf.data_ptr())
    gpu.cuda.synchronize()
    # uniform-range coefficients (motion.c:644-647) against the f64 port, all of them
    need_gb = n * (8 + 8 + 4) / 2 ** 30 * 1.3
    if free_host_gb() > need_gb + 8:
        ref = ol.r2r_many(pix_h.astype(np.float64), [d, h, w], [ol.REDFT10] * 3, impl="port", threads=host_threads())
        ol.lib().oracle_motion_uniform_f64(ref.ctypes.data, d, h, w, h, w, 1)
        got = cbuf.cpu().numpy().ravel()
        scale = np.abs(ref).max()
        err = 0.0
        for i in range(0, n, 1 << 26):                      # in slices: no second full-size temporary
            err = max(err, float(np.abs(got[i:i + (1 << 26)] - ref[i:i + (1 << 26)]).max()))
        assert err <= 1e-5 * scale, (plane, err / scale)
        del ref, got
    else:
        assert plane == "luma", "the chroma plane's reference needs 3 GB of host memory"
        # not enough host memory for the 4.2 GB f64 reference: the DC term (uniform range: u[0] = 8 d h w mean) and the roundtrip below
        mean = float(pix.double().mean())
        assert abs(float(cbuf[0, 0, 0]) / (8.0 * d * h * w) - mean) < 1e-3
    inv.execute(cbuf.data_ptr())
    out = gpu.empty_like(pix)
    assert L.dspfft_f32_to_u8(out.data_ptr(), cbuf.data_ptr(), 1.0 / (8.0 * d * h * w), n, None) == 0
    gpu.cuda.synchronize()
    assert int((out != pix).sum()) == 0
