"""GPU (-m gpu): the outer-radix-2 column split (row pairs + half tiles) through the C ABI against the oracle: the natural
cases (3840x2160 and 7680x4320 RGB), forced on smaller frames, and split vs plain passes on the same data."""
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


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def plans(h, w, c, env=None):
    from dspfun_amd import Plan, REDFT10, REDFT01
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        return Plan.image(h, w, c, REDFT10), Plan.image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * w * h))
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("h,w", [(512, 512), (1080, 1920)])
def test_forced_split_vs_oracle(gpu, h, w):
    c = 3
    x = ol.synth_f32(0xD5F0002, h * w * c).reshape(h, w, c)
    fwd, inv = plans(h, w, c, {"DSPFFT_FORCE_SPLIT": "1"})
    assert "COL*/2" in fwd.describe() and "ROW*2" in inv.describe()
    ref = ol.dct2d_interleaved(x.astype(np.float64), 5, impl="port", threads=8)
    d = dev(gpu, x)
    fwd.execute(d.data_ptr())
    got = d.cpu().numpy().astype(np.float64)
    assert np.abs(got - ref).max() / np.abs(ref).max() <= 1e-5
    inv.execute(d.data_ptr())
    assert np.abs(d.cpu().numpy() - x).max() <= 5e-6
    # out of place, inverse alone on the oracle's coefficients
    co = dev(gpu, ref.astype(np.float32))
    out = gpu.zeros_like(co)
    inv.execute(co.data_ptr(), out.data_ptr())
    assert np.abs(out.cpu().numpy() - x).max() <= 5e-6


def test_4k_forced_split_matches_plain_and_oracle(gpu):
    h, w, c = 2160, 3840, 3
    x = ol.synth_f32(0xD5F0002, h * w * c).reshape(h, w, c)
    fwd, inv = plans(h, w, c, {"DSPFFT_FORCE_SPLIT": "1"})
    pf, pi = plans(h, w, c)
    assert "COL*/2 N=2160 as 2 x 1080, K=16" in fwd.describe() and "COL*/2" not in pf.describe()   # 4K keeps the plain passes by default
    a, b = dev(gpu, x), dev(gpu, x)
    fwd.execute(a.data_ptr())
    pf.execute(b.data_ptr())
    gpu.cuda.synchronize()
    assert float((a - b).abs().max() / b.abs().max()) < 2e-6
    ref = ol.dct2d_interleaved(x.astype(np.float64), 5, impl="port", threads=8)
    got = a.cpu().numpy().astype(np.float64)
    assert np.abs(got - ref).max() / np.abs(ref).max() <= 1e-5
    assert np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)) <= 1e-5
    del ref, got
    inv.execute(a.data_ptr())
    pi.execute(b.data_ptr())
    gpu.cuda.synchronize()
    assert float((a.cpu() - gpu.from_numpy(x)).abs().max()) <= 5e-6
    assert float((b.cpu() - gpu.from_numpy(x)).abs().max()) <= 5e-6
    # per-pass execution (what bench.py times) walks the split list too
    e = dev(gpu, x)
    for i in range(fwd.num_passes):
        fwd.execute_pass(i, e.data_ptr())
    f = dev(gpu, x)
    fwd.execute(f.data_ptr())
    gpu.cuda.synchronize()
    assert bool((e == f).all())


def test_8k_split_roundtrip_and_plain_agreement(gpu):
    h, w, c = 4320, 7680, 3
    x = ol.synth_f32(0xD5F0004, h * w * c).reshape(h, w, c)
    fwd, inv = plans(h, w, c)
    pf, _ = plans(h, w, c, {"DSPFFT_NO_SPLIT": "1"})
    assert "COL*/2 N=4320 as 2 x 2160, K=16" in fwd.describe()
    a, b = dev(gpu, x), dev(gpu, x)
    fwd.execute(a.data_ptr())
    pf.execute(b.data_ptr())
    gpu.cuda.synchronize()
    assert float((a - b).abs().max() / b.abs().max()) < 2e-6
    del b
    inv.execute(a.data_ptr())
    gpu.cuda.synchronize()
    assert float((a.cpu() - gpu.from_numpy(x)).abs().max()) <= 5e-6
