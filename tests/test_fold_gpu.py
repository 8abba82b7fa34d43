"""GPU (-m gpu): folded row passes (dct_fold.h RowFoldT) through the C ABI -- the 7680 x 3 lines of BASELINE config 4's 8K frames as two
half-length transforms through half the LDS, alone and as the row pairs of the split column pass (in place: the partner workgroups'
handshake), against the f64 port and against the plain kernels (DSPFFT_FOLD=0)."""
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


def plan(h, w, c, kind, env=None):
    from dspfun_amd import Plan
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        return Plan.image(h, w, c, kind)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def relerr(got, ref):
    return np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max()


@pytest.mark.parametrize("h,w,env", [(64, 7680, {}), (48, 1920, {"DSPFFT_FOLD": "1"})])
def test_folded_rows_vs_f64_port(gpu, h, w, env):
    from dspfun_amd import REDFT10, REDFT01
    c = 3
    x = ol.synth_f32(0xD5F0004, h * w * c).reshape(h, w, c)
    fwd = plan(h, w, c, REDFT10, env)
    inv = plan(h, w, c, REDFT01, env).set_scale(1.0 / (4 * w * h))
    assert "fold#" in fwd.describe() and "fold#" in inv.describe(), fwd.describe()
    ref = ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=8)
    d = gpu.from_numpy(x).to("cuda:0")
    fwd.execute(d.data_ptr())
    assert relerr(d.cpu().numpy(), ref) < 1e-5
    inv.execute(d.data_ptr())
    assert np.abs(d.cpu().numpy() - x).max() < 5e-6
    # out of place from the port's coefficients; the input stays as it was
    co = gpu.from_numpy(ref.astype(np.float32)).to("cuda:0")
    keep = co.clone()
    out = gpu.empty_like(co)
    inv.execute(co.data_ptr(), out.data_ptr())
    assert np.abs(out.cpu().numpy() - x).max() < 5e-6 and gpu.equal(co, keep)


def test_8k_frame_folded_pairs_against_plain_kernels(gpu):
    """7680 x 4320 x 3 (BASELINE config 4's frame): the split plan with folded row pairs, in place (handshake between the partner workgroups)
    and out of place, against the same plan on the round-4 row-pair kernel; repeated, so that a race would have many chances to show"""
    from dspfun_amd import REDFT10, REDFT01
    h, w, c = 4320, 7680, 3
    x = gpu.rand(h, w, c, device="cuda:0")
    fwd, inv = plan(h, w, c, REDFT10), plan(h, w, c, REDFT01).set_scale(1.0 / (4.0 * w * h))
    fwd0, inv0 = plan(h, w, c, REDFT10, {"DSPFFT_FOLD": "0"}), plan(h, w, c, REDFT01, {"DSPFFT_FOLD": "0"}).set_scale(1.0 / (4.0 * w * h))
    assert "ROW*2" in fwd.describe() and "fold#" in fwd.describe() and "fold#" in inv.describe() and "fold#" not in fwd0.describe()
    ref = x.clone()
    fwd0.execute(ref.data_ptr())
    scale = float(ref.abs().max())
    for _ in range(5):
        d = x.clone()
        fwd.execute(d.data_ptr())
        assert float((d - ref).abs().max()) / scale < 2e-6
    out = gpu.empty_like(x)
    fwd.execute(x.data_ptr(), out.data_ptr())
    assert float((out - ref).abs().max()) / scale < 2e-6
    back = ref.clone()
    inv0.execute(back.data_ptr())
    for _ in range(5):
        d = ref.clone()
        inv.execute(d.data_ptr())
        assert float((d - back).abs().max()) < 2e-6
        assert float((d - x).abs().max()) < 5e-6
