"""GPU (-m gpu): the rest of applybasis: the general partial sums as two batched launches (forward with
offsets and partial sums, --inverse, complex `.coeff` input), the rendered frame, and the `.coeff` file round trip."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
FUNCS = ["dft", "idft", "dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4", "wht", "dht"]


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    return torch


def oracle_ex(pre, pim, func, ortho, K, N, P, off=(0, 0), inverse=False):
    h, w, _ = pre.shape
    out = np.zeros((K[1], K[0], N[1], N[0], 3, 2))
    L = ol.lib()
    L.oracle_applybasis_partsums_ex_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 10 + [C.c_longlong, C.c_longlong, C.c_int]
    a = np.ascontiguousarray(pre, dtype=np.float64)
    b = np.ascontiguousarray(pim, dtype=np.float64) if pim is not None else None
    L.oracle_applybasis_partsums_ex_f64(out.ctypes.data, a.ctypes.data, b.ctypes.data if b is not None else None, w, h, FUNCS.index(func), int(ortho),
                                        K[0], K[1], N[0], N[1], P[0], P[1], off[0], off[1], int(inverse))
    return out[..., 0] + 1j * out[..., 1]


@pytest.mark.parametrize("func", ["dft", "dct2", "dst3", "wht", "dht"])
@pytest.mark.parametrize("case", ["forward_partial", "inverse_terms", "inverse_offset", "complex_input"])
def test_partsums_general_form(gpu, func, case):
    from dspfun_amd import applybasis as ab
    w, h = 32, 16
    pre = (ol.synth_f32(3, w * h * 3).reshape(h, w, 3) * 2 - 1).astype(np.float32)
    pim = None
    if case == "forward_partial":
        K, N, P, off = (5, 3), (w // 4, h // 2), (4, 2), (1, 2)
    elif case == "inverse_terms":          # --inverse -t 24x8 -u 3x2: K = image size, N = terms / partsum (blocks cover only part of the image)
        K, N, P, off = (w, h), (8, 4), (3, 2), (0, 0)
    elif case == "inverse_offset":         # --inverse -O 2x-1: the offset moves the function's sample index, not the pixel read (applybasis.c:372-378,416-420)
        K, N, P, off = (w, h), (8, 4), (3, 2), (2, -1)
    else:                                   # a .coeff read back: complex pixels, orthogonal bases, full sums
        pim = (ol.synth_f32(4, w * h * 3).reshape(h, w, 3) - 0.5).astype(np.float32)
        K, N, P, off = (w, h), (1, 1), (w, h), (0, 0)
    ortho = case == "complex_input"
    inverse = case.startswith("inverse")
    got = ab.partsums_ex(gpu, gpu.from_numpy(pre).cuda(), gpu.from_numpy(pim).cuda() if pim is not None else None, func, ortho, K, N, P, off, inverse).cpu().numpy()
    ref = oracle_ex(pre, pim, func, ortho, K, N, P, off, inverse)
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("P", [4, 8, 16])
@pytest.mark.parametrize("func,case", [("dct2", "forward"), ("dft", "forward"), ("dst3", "ragged"), ("idft", "ragged"), ("dct3", "inverse_offset"), ("dht", "complex_pixels"),
                                       ("dft", "complex_pixels")])
def test_small_square_blocks_take_the_one_launch_kernel(gpu, monkeypatch, P, func, case):
    """-P 4x4 / 8x8 / 16x16 (applybasis.c:410-431 with small partial-sum blocks: zoom_gemm.hip ab_blocks_kernel, matrix cores for the row sums, the complex
    result written once) against the f64 restatement AND against the three-pass path it replaces (DSPFFT_AB_BLOCKS=0 in a fresh process is the A/B
    switch; here the comparison runs through the general entry with a non-square block, which never takes the kernel): real and complex bases, basis
    counts and block counts that are no multiples of 16, the --offset forms, complex pixels"""
    from dspfun_amd import applybasis as ab
    w, h = 20 * P, 7 * P
    pre = (ol.synth_f32(21 + P, w * h * 3).reshape(h, w, 3) * 2 - 1).astype(np.float32)
    pim, inverse, ortho = None, False, False
    if case == "forward":
        K, N, off = (32, 16), (w // P, h // P), (0, 0)
    elif case == "ragged":                  # 21 x 37 basis functions, 17 x 5 blocks: tiles of 16 with tails in every direction, an --offset on the basis index
        K, N, off = (21, 37), (17, 5), (3, -2)
    elif case == "inverse_offset":          # --inverse: the offset moves the function's sample index (applybasis.c:372-378,416-420)
        K, N, off, inverse = (w, h), (19, 6), (1, -1), True
    else:                                    # a .coeff input: the imaginary pixels are a second pass that adds i x its result in place
        pim = (ol.synth_f32(5 + P, w * h * 3).reshape(h, w, 3) - 0.5).astype(np.float32)
        K, N, off, ortho = (24, 40), (w // P, h // P), (0, 0), True
    dev = lambda a: gpu.from_numpy(a).cuda() if a is not None else None
    got = ab.partsums_ex(gpu, dev(pre), dev(pim), func, ortho, K, N, (P, P), off, inverse).cpu().numpy()
    ref = oracle_ex(pre, pim, func, ortho, K, N, (P, P), off, inverse)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_forward_then_inverse_through_a_coeff_file_restores_the_image(gpu, tmp_path):
    """applybasis -f dct2 -u WxH -d x.coeff img; applybasis -f dct3 -I -u WxH x.coeff: the orthogonal DCT-II spectrum written as a
    .coeff file (byte-exact round trip of the file), read back as complex pixels, inverted with DCT-III x 4/(W H) ... = the image"""
    from dspfun_amd import applybasis as ab
    w, h = 32, 16
    img = (ol.synth_f32(9, w * h * 3).reshape(h, w, 3) * 2 - 1).astype(np.float32)
    spec = ab.partsums_ex(gpu, gpu.from_numpy(img).cuda(), None, "dct2", True, (w, h), (1, 1), (w, h)).cpu().numpy()      # [h][w][1][1][3]
    path = tmp_path / "x.coeff"
    ab.write_coeff(path, spec)                                     # the default reference build's layout: complex long double (applybasis/Makefile:1-2)
    raw = open(path, "rb").read()
    assert len(raw) == 16 + w * h * 3 * 32 and np.frombuffer(raw[:16], dtype=np.uint64).tolist() == [w, h]
    back = ab.read_coeff(path)                                     # [h][w][3] complex128, the layout applybasis.c:326-331 reads
    assert np.array_equal(back, spec.reshape(h, w, 3).astype(np.complex128))
    ab.write_coeff(tmp_path / "y.coeff", back.reshape(h, w, 1, 1, 3), "D")          # an INTERMEDIATE_PRECISION=D build's file
    rawd = open(tmp_path / "y.coeff", "rb").read()
    assert len(rawd) == 16 + w * h * 3 * 16 and np.array_equal(ab.read_coeff(tmp_path / "y.coeff"), back)
    ab.write_coeff(tmp_path / "z.coeff", ab.read_coeff(tmp_path / "y.coeff").reshape(h, w, 1, 1, 3), "D")
    assert open(tmp_path / "z.coeff", "rb").read() == rawd         # byte-exact round trip
    pre, pim = np.ascontiguousarray(back.real, dtype=np.float32), np.ascontiguousarray(back.imag, dtype=np.float32)
    rec = ab.partsums_ex(gpu, gpu.from_numpy(pre).cuda(), gpu.from_numpy(pim).cuda(), "dct3", True, (w, h), (1, 1), (w, h)).cpu().numpy()
    # orthogonal DCT-II (sqrt2 on k != 0) followed by orthogonal DCT-III (x 2 on n = 0 ... sqrt2 otherwise): x (W H) / ... -> image x W H / 1
    rec = rec.reshape(h, w, 3).real / (w * h)
    assert np.abs(rec - img).max() <= 1e-4


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("plane,rescale,range_", [("real", ("linear",), "shift"), ("magnitude", ("log",), "abs"), ("phase", ("gain", "level"), "hue"),
                                                  ("imaginary", ("linear", "log"), "invert")])
def test_rendered_frame(gpu, inverse, plane, rescale, range_):
    from dspfun_amd import applybasis as ab
    w, h = 16, 8
    img = (ol.synth_f32(21, w * h * 3).reshape(h, w, 3) * 2 - 1).astype(np.float32)
    K, N, P = ((w, h), (4, 2), (2, 2)) if inverse else ((3, 2), (w // 2, h // 2), (2, 2))
    parts = ab.partsums_ex(gpu, gpu.from_numpy(img).cuda(), None, "dft", False, K, N, P)
    scale, padding = 2, 1
    coeff_scale = float(P[0] * P[1])
    frame = ab.render(gpu, parts, inverse, scale, padding, plane, rescale, range_, coeff_scale, float(w * h), padcolor=(0.25, 0.5, 0.75, 1.0)).cpu().numpy()
    ref = np.empty(frame.shape, dtype=np.float64)
    ref[...] = (0.25, 0.5, 0.75, 1.0)
    pn = parts.cpu().numpy()
    flat = np.ascontiguousarray(np.stack([pn.real, pn.imag], axis=-1), dtype=np.float64)
    L = ol.lib()
    L.oracle_applybasis_render_f64.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 11 + [C.c_double, C.c_double]
    r1 = ab.RESCALES.index(rescale[1]) if len(rescale) > 1 else -1
    L.oracle_applybasis_render_f64(ref.ctypes.data, flat.ctypes.data, K[0], K[1], N[0], N[1], int(inverse), scale, padding, ab.PLANES.index(plane),
                                   ab.RESCALES.index(rescale[0]), r1, ab.RANGES.index(range_), coeff_scale, float(w * h))
    assert frame.shape == ref.shape
    assert np.abs(frame - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
