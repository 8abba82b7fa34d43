"""CPU: folded row passes (dct_fold.h RowFoldT, spec_kernels.h row_fold_kernel) on the test-only emulation backend, against the oracle.
A line of N samples runs as a half-length REDFT10 / REDFT01 of its mirror sums and a DCT-IV of its mirror differences through half the
LDS.  DSPFFT_FOLD=1 takes the folded kernel wherever spec_list.h lists one (1920 x 3: small enough for the emulation); the natural
case, 7680 x 3 lines, is checked here on a few lines and at full size in the GPU tests."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from dspfun_amd.engine import Plan, REDFT10, REDFT01
from emul_lib import emul

TOL = 2e-6


@pytest.fixture()
def folded(monkeypatch):
    monkeypatch.setenv("DSPFFT_FOLD", "1")


def relerr(got, ref):
    return np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max()


def aligned(shape):
    """float32 array on a 64-byte boundary (the folded kernels move 16-byte pieces)"""
    n = int(np.prod(shape))
    raw = np.zeros(n + 16, dtype=np.float32)
    off = (-raw.ctypes.data % 64) // 4
    return raw[off:off + n].reshape(shape)


@pytest.mark.parametrize("h,w", [(8, 1920), (6, 7680)])
def test_folded_rows_forward_inverse_vs_oracle(folded, h, w):
    c = 3
    x = aligned((h, w, c)); x[...] = ol.synth_f32(0xD5F0002, h * w * c).reshape(h, w, c)
    fwd = Plan.image(h, w, c, REDFT10, lib=emul())
    inv = Plan.image(h, w, c, REDFT01, lib=emul()).set_scale(1.0 / (4 * w * h))
    assert "fold#" in fwd.describe() and "fold#" in inv.describe(), fwd.describe()
    ref = ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=4)
    d = aligned(x.shape); d[...] = x
    fwd.execute(d.ctypes.data)
    assert relerr(d, ref) < TOL
    inv.execute(d.ctypes.data)
    assert np.abs(d - x).max() < 5e-6
    # out of place, input untouched
    co = aligned(x.shape); co[...] = ref.astype(np.float32)
    keep = co.copy()
    out = aligned(x.shape)
    inv.execute(co.ctypes.data, out.ctypes.data)
    assert np.abs(out - x).max() < 5e-6 and np.array_equal(co, keep)


def test_folded_and_plain_rows_agree_with_scales(folded, monkeypatch):
    """per-index scales of the row axis (spec/spec.c:70-78, spec/ispec.c:153-159) ride on the folded pass's first / last orbit"""
    h, w, c = 8, 1920, 3
    x = aligned((h, w, c)); x[...] = ol.synth_f32(11, h * w * c).reshape(h, w, c)
    for kind in (REDFT10, REDFT01):
        f = Plan.image(h, w, c, kind, lib=emul()).set_scale(0.37)
        monkeypatch.setenv("DSPFFT_FOLD", "0")
        p = Plan.image(h, w, c, kind, lib=emul()).set_scale(0.37)
        monkeypatch.setenv("DSPFFT_FOLD", "1")
        assert "fold#" in f.describe() and "fold#" not in p.describe()
        for q in (f, p):
            q.set_axis_scale0(1, 0.5, 0.25)
        a = aligned(x.shape); b = aligned(x.shape); a[...] = x; b[...] = x
        f.execute(a.ctypes.data); p.execute(b.ctypes.data)
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-6
        # unaligned buffers: the folded kernel declines, the plain one runs, same numbers
        raw = np.zeros(x.size + 1, dtype=np.float32)
        u = raw[1:].reshape(x.shape) if raw.ctypes.data % 16 == 0 else raw[:-1].reshape(x.shape)
        if u.ctypes.data % 16:
            u[...] = x
            f.execute(u.ctypes.data)
            assert np.array_equal(u, b)


def test_fold_is_taken_where_the_plain_line_fills_a_cu(monkeypatch):
    monkeypatch.delenv("DSPFFT_FOLD", raising=False)
    assert "fold#" in Plan.image(4320, 7680, 3, REDFT10, lib=emul()).describe()
    assert "fold#" not in Plan.image(1080, 1920, 3, REDFT10, lib=emul()).describe()
    assert "fold#" not in Plan.image(2160, 3840, 3, REDFT10, lib=emul()).describe()
    monkeypatch.setenv("DSPFFT_FOLD", "0")
    assert "fold#" not in Plan.image(4320, 7680, 3, REDFT10, lib=emul()).describe()


def test_folded_row_pairs_of_a_split_plan(folded, monkeypatch):
    """the row-pair butterfly of the split column pass (engine.cpp build_split) with one output line per workgroup; in place the partners
    read each other's output lines"""
    monkeypatch.setenv("DSPFFT_FORCE_SPLIT", "1")
    h, w, c = 512, 1920, 3
    x = aligned((h, w, c)); x[...] = ol.synth_f32(0xD5F0004, h * w * c).reshape(h, w, c)
    fwd = Plan.image(h, w, c, REDFT10, lib=emul())
    inv = Plan.image(h, w, c, REDFT01, lib=emul()).set_scale(1.0 / (4 * w * h))
    df, di = fwd.describe().splitlines(), inv.describe().splitlines()
    assert df[1].startswith("axis 1: ROW*2") and "fold#" in df[1] and df[2].startswith("axis 0: COL*/2"), df
    assert di[1].startswith("axis 0: COL*/2") and di[2].startswith("axis 1: ROW*2") and "fold#" in di[2], di
    ref = ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=4)
    d = aligned(x.shape); d[...] = x
    fwd.execute(d.ctypes.data)
    assert relerr(d, ref) < TOL
    inv.execute(d.ctypes.data)
    assert np.abs(d - x).max() < 5e-6
    out = aligned(x.shape)
    fwd.execute(x.ctypes.data, out.ctypes.data)
    assert relerr(out, ref) < TOL


def test_folded_row_pairs_carry_the_fused_scan_step(folded, monkeypatch):
    """scan/scan.c:429-459: mask on the half-tile column pass's loads (with skipped tiles read as zeros by the row pass), accumulation in
    the folded row-pair pass's stores -- the same sums as the plain passes"""
    monkeypatch.setenv("DSPFFT_FORCE_SPLIT", "1")
    h, w, c = 512, 1920, 3
    x = aligned((h, w, c)); x[...] = ol.synth_f32(0xD5F0004, h * w * c).reshape(h, w, c)
    fwd = Plan.image(h, w, c, REDFT10, lib=emul()).set_scale(1.0 / (4 * w * h))
    inv = Plan.image(h, w, c, REDFT01, lib=emul())
    assert "fold#" in inv.describe() and "COL*/2" in inv.describe()
    monkeypatch.setenv("DSPFFT_NO_SPLIT", "1")
    monkeypatch.setenv("DSPFFT_FOLD", "0")
    inv_plain = Plan.image(h, w, c, REDFT01, lib=emul())
    assert "fold#" not in inv_plain.describe() and "COL*/2" not in inv_plain.describe()
    coeffs = aligned(x.shape); coeffs[...] = x
    fwd.execute(coeffs.ctypes.data)
    L = emul()
    nframes = 4
    step = (w * h + nframes - 1) // nframes
    ids = np.zeros(w * h, dtype=np.uint32)
    assert L.dspfft_scan_zigzag_frame_ids(ids.ctypes.data, w, h, step, None) == 0
    inv.scan_prepare(ids.ctypes.data, c)
    acc = aligned(x.shape); acc2 = aligned(x.shape); work = aligned(x.shape)
    assert L.dspfft_broadcast_dc(acc.ctypes.data, coeffs.ctypes.data, w * h, c, None) == 0
    acc2[...] = acc
    for f in range(nframes):
        inv.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc.ctypes.data, ids.ctypes.data, f, c)
        inv_plain.execute_masked_accumulate(coeffs.ctypes.data, work.ctypes.data, acc2.ctypes.data, ids.ctypes.data, f, c)
        assert np.abs(acc - acc2).max() < 2e-6, f
    assert np.abs(acc - x).max() <= 5e-6
