"""GPU (-m gpu): dspfft_gemm_nt_f32 -- the dense product behind zoom's separable basis product (zoom/zoom.c:361-375) and applybasis'
partial sums (applybasis/applybasis.c:410-431) -- through the C ABI on every kernel it dispatches to: the LDS-DMA kernel on 128 x 128,
128 x 192 and 64 x 64 tiles, the register-staged kernel it falls back to (K < 32, unaligned operands), with ragged edges in M, N and K,
padded leading dimensions, batches and the strided store.  A floating-point kernel: the reference is the same product in float64
(numpy on the host), the bound is the f32 accumulation's: 4e-7 x sum_k |a| |b| (K eps with some room)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from dspfun_amd import _lib
    _lib.load()
    return torch


def run(gpu, M, N, K, lda=None, ldb=None, ldc=None, cs=1, batch=1, alpha=1.0, misalign=0, seed=0):
    from dspfun_amd import _lib
    L = _lib.load()
    lda = lda or K; ldb = ldb or K; ldc = ldc or N * cs
    rng = np.random.default_rng(seed)
    A = (rng.random((batch, M, lda), dtype=np.float32) * 2 - 1)
    B = (rng.random((batch, N, ldb), dtype=np.float32) * 2 - 1)
    dA = gpu.zeros(A.size + 8, dtype=gpu.float32, device="cuda:0"); dB = gpu.zeros(B.size + 8, dtype=gpu.float32, device="cuda:0")
    a = dA[misalign:misalign + A.size]; b = dB[misalign:misalign + B.size]
    a.copy_(gpu.from_numpy(A.ravel())); b.copy_(gpu.from_numpy(B.ravel()))
    sc = M * ldc
    c = gpu.full((batch * sc,), float("nan"), dtype=gpu.float32, device="cuda:0")
    rc = L.dspfft_gemm_nt_f32(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, lda, ldb, ldc, cs, batch, M * lda, N * ldb, sc, alpha, None)
    assert rc == 0, L.dspfft_zoom_last_error()
    gpu.cuda.synchronize()
    got = c.cpu().numpy().reshape(batch, M, ldc)
    A64 = A[:, :, :K].astype(np.float64); B64 = B[:, :, :K].astype(np.float64)
    ref = alpha * np.einsum("bmk,bnk->bmn", A64, B64)
    bound = 4e-7 * abs(alpha) * np.einsum("bmk,bnk->bmn", np.abs(A64), np.abs(B64)) + 1e-30
    out = got[:, :, :N * cs:cs]
    assert np.isfinite(out).all()
    assert (np.abs(out - ref) <= bound).all(), float((np.abs(out - ref) / bound).max())
    if cs > 1 or ldc > N * cs:          # nothing but the addressed elements was written
        mask = np.ones(got.shape, dtype=bool); mask[:, :, :N * cs:cs] = False
        assert np.isnan(got[mask]).all()


@pytest.mark.parametrize("M,N,K", [
    (128, 128, 32), (64, 64, 64), (300, 200, 96), (130, 130, 33), (1000, 700, 100),      # 64 x 64 tiles, ragged edges, K tails
    (2048, 2048, 256), (4320, 1000, 1080), (2100, 2050, 517),                            # 128 x 128 tiles
    (7680, 3240, 64), (1280, 1152 * 3, 40),                                              # 128 x 192 where it fills the rounds
    (96, 80, 8), (200, 100, 20), (70, 50, 3),                                            # K < 32: the register-staged kernel
])
def test_product_against_float64(gpu, M, N, K):
    run(gpu, M, N, K, seed=M + N + K)


def test_padded_leading_dimensions_and_scale(gpu):
    run(gpu, 513, 770, 200, lda=256, ldb=204, ldc=800, alpha=0.125, seed=1)
    run(gpu, 2000, 1800, 96, lda=100, ldb=128, ldc=1800, alpha=-3.0, seed=2)


def test_strided_store_and_batches(gpu):
    # zoom's interleaved colour channels as round 3 stored them (cs = 3), three batches; and batches of plain products
    run(gpu, 700, 600, 160, cs=3, batch=1, seed=3)
    run(gpu, 300, 260, 64, cs=2, batch=3, seed=4)
    run(gpu, 1500, 1400, 96, batch=3, seed=5)


def test_unaligned_operands_take_the_fallback(gpu):
    # operands 4 bytes off a 16-byte boundary, a leading dimension that is no multiple of 4: same results from the register-staged kernel
    run(gpu, 400, 300, 128, misalign=1, seed=6)
    run(gpu, 400, 300, 130, lda=131, ldb=133, seed=7)


def test_k_tail_inside_aligned_padded_rows(gpu):
    # K no multiple of 4 while the rows are 16-byte aligned and padded with NON-ZERO values (run() fills the whole lda / ldb): the LDS-DMA
    # kernel masks its K tail per 16-byte piece, so such a product must not reach it (columns K .. of both operands would be summed too)
    run(gpu, 400, 300, 130, lda=132, ldb=136, seed=8)
    run(gpu, 200, 260, 35, lda=64, ldb=64, seed=9)
    run(gpu, 2048, 2048, 257, lda=260, ldb=260, seed=10)


def test_bad_arguments_are_refused(gpu):
    from dspfun_amd import _lib
    L = _lib.load()
    x = gpu.zeros(64, dtype=gpu.float32, device="cuda:0")
    assert L.dspfft_gemm_nt_f32(None, x.data_ptr(), x.data_ptr(), 4, 4, 4, 4, 4, 4, 1, 1, 0, 0, 0, 1.0, None) != 0
    assert L.dspfft_gemm_nt_f32(x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, 4, 4, 4, 4, 4, 1, 1, 0, 0, 0, 1.0, None) != 0
    assert L.dspfft_gemm_nt_f32(x.data_ptr(), x.data_ptr(), x.data_ptr(), 4, 4, 4, 4, 4, 4, 0, 1, 0, 0, 0, 1.0, None) != 0
