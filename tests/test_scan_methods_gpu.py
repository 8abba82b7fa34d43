"""GPU (-m gpu): the scan-method generators of the HIP library, bit-exact against host/scan_orders.c (same checks as the CPU
emulation run), magnitude against its numpy restatement at 1080p, and scan's whole frame loop device-resident for a single-owner
method (diagonal), for box (stamps, with pixels shared between frames) and for magnitude, against the f64 oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
import scan_device_checks as sd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dspfun_amd import _lib
    return torch, _lib.load()


@pytest.fixture(scope="module")
def so():
    return sd.host_lib()


def _mk(torch):
    def alloc(n):
        t = torch.zeros(max(1, n), dtype=torch.int32, device="cuda:0")
        return t, t.data_ptr()

    def fetch(t):
        torch.cuda.synchronize()
        return t.cpu().numpy().view(np.uint32).copy()
    return alloc, fetch


@pytest.mark.parametrize("w,h", sd.SIZES + [(96, 54)])
@pytest.mark.parametrize("m", range(len(sd.METHODS)))
def test_generators_match_the_host_library(gpu, so, m, w, h):
    torch, L = gpu
    alloc, fetch = _mk(torch)
    sd.check_method(L, so, m, w, h, alloc, fetch)


def test_radial_owner_ids_full_hd_match_libm(gpu):
    """1920x1080: the exact integer rint(hypot) against numpy's hypot + rint (glibc), every pixel"""
    torch, L = gpu
    w, h = 1920, 1080
    for m, name in ((9, "radial"), (10, "iradial")):
        t = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
        assert L.dspfft_scan_owner_index(t.data_ptr(), m, w, h, None) == 0
        ys, xs = np.divmod(np.arange(w * h), w)
        if name == "radial":
            want = np.rint(np.hypot(xs, ys))
        else:
            want = np.rint(np.hypot(w - 1, h - 1)) + 1 - np.rint(np.hypot(w - xs - 1, h - ys - 1)) - 1
        assert np.array_equal(t.cpu().numpy().astype(np.int64), want.astype(np.int64))
        assert L.dspfft_scan_limit(m, w, h) == int(want.max()) + 1


@pytest.mark.parametrize("q", [0.0, 1000.0])
def test_magnitude_index_1080p(gpu, q):
    torch, L = gpu
    from dspfun_amd import Plan, REDFT10
    w, h, ch = 1920, 1080, 3
    x = ol.synth_f32(0xD5F0003, w * h * ch).reshape(h, w, ch)
    d = torch.from_numpy(x).to("cuda:0")
    Plan.image(h, w, ch, REDFT10).set_scale(1.0 / (4.0 * w * h)).execute(d.data_ptr())
    idx = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
    work = torch.zeros(L.dspfft_scan_magnitude_work_bytes(w, h), dtype=torch.uint8, device="cuda:0")
    lim = C.c_uint32()
    assert L.dspfft_scan_magnitude_index(idx.data_ptr(), d.data_ptr(), w, h, ch, q, work.data_ptr(), work.numel(), C.byref(lim), None) == 0
    want, wlim = sd.magnitude_reference(d.cpu().numpy(), w, h, ch, q)
    assert lim.value == wlim
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), want)


@pytest.mark.parametrize("method", ["diagonal", "box", "magnitude", "mirror"])
def test_device_resident_frame_loop(gpu, so, method):
    """scan/scan.c:377-383,421-459 with every buffer on the device: per frame the mask (owner ids, or stamps for box), then ONE fused
    masked-accumulate execution; the per-frame sums against the f64 restatement of the host loop, duplicates included"""
    torch, L = gpu
    from dspfun_amd import Plan, REDFT10, REDFT01
    w, h, c = 160, 90, 3                     # wide: box re-emits row 89 for every index >= 89
    x = ol.synth_f32(77, w * h * c).reshape(h, w, c)
    coeffs = torch.from_numpy(x).to("cuda:0")
    Plan.image(h, w, c, REDFT10).set_scale(1.0 / (4.0 * w * h)).execute(coeffs.data_ptr())
    inv = Plan.image(h, w, c, REDFT01)
    cf64 = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port"))
    ol.lib().oracle_scan_normalise_f64(cf64.ctypes.data, w, h, c)
    ids = torch.full((w * h,), -1, dtype=torch.int32, device="cuda:0")
    if method == "magnitude":
        idx = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
        work = torch.zeros(L.dspfft_scan_magnitude_work_bytes(w, h), dtype=torch.uint8, device="cuda:0")
        lim = C.c_uint32()
        assert L.dspfft_scan_magnitude_index(idx.data_ptr(), coeffs.data_ptr(), w, h, c, 0.0, work.data_ptr(), work.numel(), C.byref(lim), None) == 0
        limit = lim.value
        owner = idx.cpu().numpy().astype(np.int64)
        lists = [np.flatnonzero(owner == i) for i in range(limit)] if limit < 4000 else None
    else:
        m = sd.METHODS.index(method)
        limit = L.dspfft_scan_limit(m, w, h)
        lists = [np.array([y * w + xx for (y, xx) in cs if y * w + xx < w * h], dtype=np.int64) for cs in sd.host_orders(so, m, w, h)]
    nframes = 6
    step = (limit + nframes - 1) // nframes
    if method == "magnitude":
        ids.copy_(idx)
        assert L.dspfft_scan_index_to_frame_ids(ids.data_ptr(), w * h, step, None) == 0
    elif method != "box":
        assert L.dspfft_scan_frame_ids(ids.data_ptr(), m, w, h, step, None) == 0
    else:
        slots = L.dspfft_scan_coord_slots(m, w, h)
        lin = torch.zeros(step * slots, dtype=torch.int32, device="cuda:0")
    acc = torch.empty_like(coeffs)
    work2 = torch.empty_like(coeffs)
    assert L.dspfft_broadcast_dc(acc.data_ptr(), coeffs.data_ptr(), w * h, c, None) == 0
    ref = np.ascontiguousarray(np.broadcast_to(cf64[0, 0], (h, w, c)).copy())
    for f in range(nframes):
        lo, hi = f * step, min(limit, (f + 1) * step)
        if method == "box":
            assert L.dspfft_scan_coords(lin.data_ptr(), m, w, h, lo, hi - lo, None) == 0
            assert L.dspfft_scan_stamp(ids.data_ptr(), lin.data_ptr(), (hi - lo) * slots, f, None) == 0
        inv.execute_masked_accumulate(coeffs.data_ptr(), work2.data_ptr(), acc.data_ptr(), ids.data_ptr(), f, c)
        sel = np.zeros(w * h, dtype=bool)
        if lists is not None:
            for i in range(lo, hi):
                sel[lists[i]] = True
        else:
            sel = (owner // step) == f
        sel[0] = False
        rec = np.where(sel.reshape(h, w)[:, :, None], cf64, 0.0)
        ref += ol.dct2d_interleaved(rec, REDFT01, impl="port")
        torch.cuda.synchronize()
        assert np.abs(acc.cpu().numpy() - ref).max() < 5e-6, (method, f)
    if method != "box":                      # a partition of the pixels: the frames add up to the input (box counts shared pixels twice)
        assert np.abs(acc.cpu().numpy() - x).max() <= 5e-6
