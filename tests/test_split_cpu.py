"""CPU: the outer-radix-2 split of a long column axis (dct_spec.h ColHalfSpec + row_pair_kernel; engine.cpp build_split) on the
test-only emulation backend, against the oracle.  DSPFFT_FORCE_SPLIT=1 applies the split wherever the kernels exist, so the
mechanism is checked on frames small enough for the emulation; the natural case (3840x2160, 7680x4320) runs in the GPU tests."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from dspfun_amd.engine import Plan, REDFT10, REDFT01
from emul_lib import emul

TOL = 2e-6


@pytest.fixture()
def forced():
    os.environ["DSPFFT_FORCE_SPLIT"] = "1"
    yield
    del os.environ["DSPFFT_FORCE_SPLIT"]


def relerr(got, ref):
    return np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max()


@pytest.mark.parametrize("pipe", ["0", "1"])       # the pair's butterfly on the input side (row_pair_kernel) / on the output side (row_pair_pipe_kernel's order of work)
@pytest.mark.parametrize("h,w", [(512, 512), (1080, 1920)])
def test_forced_split_forward_inverse_vs_oracle(forced, h, w, pipe, monkeypatch):
    monkeypatch.setenv("EMUL_PAIR_PIPE", pipe)
    c = 3
    x = ol.synth_f32(0xD5F0002, h * w * c).reshape(h, w, c)
    fwd = Plan.image(h, w, c, REDFT10, lib=emul())
    inv = Plan.image(h, w, c, REDFT01, lib=emul()).set_scale(1.0 / (4 * w * h))
    df, di = fwd.describe().splitlines(), inv.describe().splitlines()
    # REDFT10 runs row pairs first, REDFT01 half tiles first; the plain passes stay available
    assert df[1].startswith("axis 1: ROW*2") and df[2].startswith("axis 0: COL*/2") and df[3].startswith("plain axis 1"), df
    assert di[1].startswith("axis 0: COL*/2") and di[2].startswith("axis 1: ROW*2"), di
    ref = ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=4)
    d = x.copy()
    fwd.execute(d.ctypes.data)
    assert relerr(d, ref) < TOL
    inv.execute(d.ctypes.data)
    assert np.abs(d - x).max() < 5e-6
    # the inverse alone on the oracle's coefficients, out of place
    co = ref.astype(np.float32)
    out = np.zeros_like(co)
    keep = co.copy()
    inv.execute(co.ctypes.data, out.ctypes.data)
    assert np.abs(out - x).max() < 5e-6 and np.array_equal(co, keep)


@pytest.mark.parametrize("pipe", ["0", None])
def test_forced_split_on_8k_wide_lines(forced, pipe, monkeypatch):
    """7680 x 3 lines (BASELINE config 4's rows) as row pairs of a split plan: the 1024-thread pair kernel's phases on the emulation, and (default, as the
    HIP launcher chooses for plain passes) the pipelined pair kernel's order of work on its 768 threads: T(r1) held as values, T(r2) added and
    subtracted in the closing phase (RowSpecG::final_each)"""
    if pipe is not None:
        monkeypatch.setenv("EMUL_PAIR_PIPE", pipe)
    h, w, c = 512, 7680, 3
    x = ol.synth_f32(0xD5F0004, h * w * c).reshape(h, w, c)
    fwd = Plan.image(h, w, c, REDFT10, lib=emul())
    inv = Plan.image(h, w, c, REDFT01, lib=emul()).set_scale(1.0 / (4 * w * h))
    assert "ROW*2 N=7680" in fwd.describe() and "ROW*2 N=7680" in inv.describe()
    ref = ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=4)
    d = x.copy()
    fwd.execute(d.ctypes.data)
    assert relerr(d, ref) < TOL
    inv.execute(d.ctypes.data)
    assert np.abs(d - x).max() < 5e-6


def test_split_and_plain_agree_and_fallbacks(forced):
    h, w, c = 512, 512, 3
    x = ol.synth_f32(7, h * w * c).reshape(h, w, c)
    sp = Plan.image(h, w, c, REDFT10, lib=emul())
    os.environ["DSPFFT_NO_SPLIT"] = "1"
    try:
        pl = Plan.image(h, w, c, REDFT10, lib=emul())
    finally:
        del os.environ["DSPFFT_NO_SPLIT"]
    assert "COL*/2" in sp.describe() and "COL*/2" not in pl.describe()
    a, b = x.copy(), x.copy()
    sp.execute(a.ctypes.data)
    pl.execute(b.ctypes.data)
    assert np.abs(a - b).max() / np.abs(b).max() < 1e-6
    # a scale on index 0 of the split axis' INPUT (REDFT10) cannot ride on the row-pair butterfly: the plain passes run
    sp.set_axis_scale0(0, 0.5, 1.0)
    pl.set_axis_scale0(0, 0.5, 1.0)
    a, b = x.copy(), x.copy()
    sp.execute(a.ctypes.data)
    pl.execute(b.ctypes.data)
    assert np.array_equal(a, b)
    # per-pass execution walks the same list as execute()
    sp.set_axis_scale0(0, 1.0, 0.25)
    pl.set_axis_scale0(0, 1.0, 0.25)
    a, b = x.copy(), x.copy()
    for i in range(sp.num_passes):
        sp.execute_pass(i, a.ctypes.data)
    pl.execute(b.ctypes.data)
    assert np.abs(a - b).max() / np.abs(b).max() < 1e-6


def test_split_is_not_used_where_it_does_not_apply():
    # default policy: only when the full-length column tile fills a CU's LDS on its own
    assert "COL*/2" not in Plan.image(512, 512, 3, REDFT10, lib=emul()).describe()
    # 4K: wider half tiles exist but measured no faster with the paired row pass (engine.cpp build_split): plain passes
    assert "COL*/2" not in Plan.image(2160, 3840, 3, REDFT10, lib=emul()).describe()
    d = Plan.image(4320, 7680, 3, REDFT01, lib=emul()).describe()
    assert d.splitlines()[1].startswith("axis 0: COL*/2 N=4320 as 2 x 2160, K=16")


@pytest.mark.parametrize("pipe", ["0", "1"])       # row_pair_kernel's order of work / the pipelined pair kernel's scan form (loads by precomputed tile flags, accumulation in r2's closing phase)
def test_forced_split_carries_the_fused_scan_step(forced, pipe, monkeypatch):
    """scan/scan.c:429-459 through the split passes: mask on the half-tile column pass's loads, accumulation in the row-pair pass's
    stores -- the same sums as the plain passes and as the f64 restatement"""
    monkeypatch.setenv("EMUL_PAIR_PIPE", pipe)
    h, w, c = 512, 512, 3
    x = ol.synth_f32(0xD5F0004, h * w * c).reshape(h, w, c)
    fwd = Plan.image(h, w, c, REDFT10, lib=emul()).set_scale(1.0 / (4 * w * h))
    inv = Plan.image(h, w, c, REDFT01, lib=emul())
    assert "COL*/2" in inv.describe()
    os.environ["DSPFFT_NO_SPLIT"] = "1"
    try:
        inv_plain = Plan.image(h, w, c, REDFT01, lib=emul())
    finally:
        del os.environ["DSPFFT_NO_SPLIT"]
    coeffs = x.copy()
    fwd.execute(coeffs.ctypes.data)
    L = emul()
    step = (w * h + 3) // 4
    ids = np.zeros(w * h, dtype=np.uint32)
    assert L.dspfft_scan_zigzag_frame_ids(ids.ctypes.data, w, h, step, None) == 0
    acc = np.zeros_like(x); acc2 = np.zeros_like(x); work = np.zeros_like(x)
    assert L.dspfft_broadcast_dc(acc.ctypes.data, coeffs.ctypes.data, w * h, c, None) == 0
    acc2[...] = acc
    for f in range(4):
        inv.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc.ctypes.data, ids.ctypes.data, f, c)
        inv_plain.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc2.ctypes.data, ids.ctypes.data, f, c)
        assert np.abs(acc - acc2).max() < 2e-6, f
    assert np.abs(acc - x).max() <= 5e-6


# ---- sparse scan frames: the masked column pass skips tiles without selected coefficients, the row pass reads zeros there ----
@pytest.mark.parametrize("force_split", [False, True])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_masked_accumulate_skips_empty_tiles(force_split, dtype, monkeypatch):
    import ctypes as C
    h, w, c = 512, 512, 3
    if force_split and dtype == "f64":
        pytest.skip("double plans have no column split")
    if force_split:
        monkeypatch.setenv("DSPFFT_FORCE_SPLIT", "1")
        monkeypatch.setenv("EMUL_PAIR_PIPE", "1")     # the pair pass in the pipelined kernel's scan form: the skipped tiles through RowSpecG::flag_bits01 / prefetch01_bits
    else:
        monkeypatch.setenv("DSPFFT_ZSKIP", "1")       # plain plans take part on request only
    L = emul()
    npdt = np.float32 if dtype == "f32" else np.float64
    x = ol.synth_f32(4242, h * w * c).astype(npdt).reshape(h, w, c)
    coeffs = x.copy()
    Plan.image(h, w, c, REDFT10, lib=L, dtype=dtype).set_scale(1.0 / (4 * w * h)).execute(coeffs.ctypes.data)
    nframes = 7
    ids = np.zeros(h * w, dtype=np.uint32)
    assert L.dspfft_scan_zigzag_frame_ids(ids.ctypes.data, w, h, (h * w + nframes - 1) // nframes, None) == 0

    def aligned(shape):          # the double kernels move 32-byte lane vectors
        raw = np.empty(int(np.prod(shape)) * np.dtype(npdt).itemsize + 64, dtype=np.uint8)
        o = (-raw.ctypes.data) % 64
        return raw[o:o + int(np.prod(shape)) * np.dtype(npdt).itemsize].view(npdt).reshape(shape)
    cal = aligned(coeffs.shape); cal[...] = coeffs; coeffs = cal

    def run(skip, prepare=False):
        if skip:
            monkeypatch.delenv("DSPFFT_NO_ZSKIP", raising=False)
        else:
            monkeypatch.setenv("DSPFFT_NO_ZSKIP", "1")
        inv = Plan.image(h, w, c, REDFT01, lib=L, dtype=dtype)
        if prepare:
            inv.scan_prepare(ids.ctypes.data, c)
        assert ("ROW*2" in inv.describe()) == force_split
        acc = aligned((h, w, c)); acc[...] = coeffs[0, 0]
        work = aligned((h, w, c))
        sums = []
        for f in range(nframes):
            work[...] = np.nan            # a skipped tile must not be read back
            inv.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc.ctypes.data, ids.ctypes.data, f, c)
            sums.append(acc.copy())
            if skip and f in (0, nframes - 1):
                # the first and the last zigzag frame touch a corner of the spectrum only: most column tiles stayed untouched
                assert np.isnan(work).mean() > 0.3
        return sums
    a, b, p = run(True), run(False), run(True, prepare=True)
    for sa, sp in zip(a, p):
        assert np.array_equal(sa, sp)              # prepared id ranges only decide earlier what the kernel finds out anyway
    for sa, sb in zip(a, b):
        if force_split:
            assert np.array_equal(sa, sb)          # same passes in the same order: skipping a tile of zeros changes nothing
        else:
            # the plain plan runs its column pass first when it skips (the passes commute): equal to rounding
            assert np.abs(sa - sb).max() < (2e-6 if dtype == "f32" else 1e-14)
    assert np.abs(a[-1] - x).max() < (2e-5 if dtype == "f32" else 1e-12)


@pytest.mark.parametrize("force_split", [False, True])
@pytest.mark.parametrize("nframes", [7, 300])
def test_masked_accumulate_prepared_id_table(force_split, nframes, monkeypatch):
    """dspfft_plan_scan_prepare also lays the owner ids out in the column tiles' reading order (one byte per id below 255 frames, two
    above: dct_spec.h masked_two_step); a frame step with the table is bit for bit the step that reads the id array itself, and a frame
    id beyond the table's width falls back to the array"""
    h, w, c = 512, 512, 3
    if force_split:
        monkeypatch.setenv("DSPFFT_FORCE_SPLIT", "1")
    else:
        monkeypatch.setenv("DSPFFT_ZSKIP", "1")
    L = emul()
    x = ol.synth_f32(777, h * w * c).reshape(h, w, c)
    coeffs = x.copy()
    Plan.image(h, w, c, REDFT10, lib=L).set_scale(1.0 / (4 * w * h)).execute(coeffs.ctypes.data)
    ids = np.zeros(h * w, dtype=np.uint32)
    assert L.dspfft_scan_zigzag_frame_ids(ids.ctypes.data, w, h, (h * w + nframes - 1) // nframes, None) == 0
    assert int(ids[ids != 0xFFFFFFFF].max()) == nframes - 1
    frames = [0, nframes // 2, nframes - 1, nframes + 5]          # the last one is nobody's: adds zeros

    def run(prepare, eids):
        monkeypatch.setenv("DSPFFT_SCAN_EIDS", "1" if eids else "0")
        inv = Plan.image(h, w, c, REDFT01, lib=emul())
        if prepare:
            inv.scan_prepare(ids.ctypes.data, c)
        acc = np.zeros((h, w, c), dtype=np.float32)
        work = np.empty((h, w, c), dtype=np.float32)
        out = []
        for f in frames:
            inv.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc.ctypes.data, ids.ctypes.data, f, c)
            out.append(acc.copy())
        return out
    with_table, without, unprepared = run(True, True), run(True, False), run(False, False)
    for a, b, u in zip(with_table, without, unprepared):
        assert np.array_equal(a, b) and np.array_equal(a, u)
    assert np.array_equal(with_table[-1], with_table[-2])      # the frame nobody owns changed nothing
