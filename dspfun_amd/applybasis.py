"""Host-side mirror of applybasis' forward partial sums (applybasis/applybasis.c:392-448, the arithmetic
behind both the rendered frame and the `.coeff` dump) over device memory."""
import ctypes as C

from . import _lib
from .engine import DspfftError

FUNCTIONS = ("dft", "idft", "dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4", "wht", "dht")   # applybasis.c:77-140


def partsums(torch, pixels_hwc, function="dft", orthogonal=False, terms=None, partsum=(1, 1), offset=(0, 0)):
    """pixels_hwc: (h, w, 3) f32 cuda tensor, already range-mapped (applybasis.c:358-360).
    terms/partsum/offset are (w, h) pairs as on the tool's command line (-t, -u, -O).
    Returns a complex64 tensor [Kh, Kw, Nh, Nw, 3]."""
    lib = _lib.load()
    h, w, c = pixels_hwc.shape
    assert c == 3 and pixels_hwc.dtype == torch.float32 and pixels_hwc.is_cuda and pixels_hwc.is_contiguous()
    func = FUNCTIONS.index(function)
    kw, kh = terms if terms else (w, h)
    pw, ph = partsum
    nw, nh = w // pw, h // ph
    out = torch.empty((kh, kw, nh, nw, 3, 2), dtype=torch.float32, device=pixels_hwc.device)
    work = torch.empty(lib.dspfft_applybasis_work_floats(w, h, kw, kh, pw, ph, func), dtype=torch.float32, device=pixels_hwc.device)
    rc = lib.dspfft_applybasis_partsums(out.data_ptr(), pixels_hwc.data_ptr(), w, h, func, int(orthogonal), kw, kh, pw, ph,
                                        offset[0], offset[1], work.data_ptr(), None)
    if rc:
        raise DspfftError(lib.dspfft_zoom_last_error().decode())
    return torch.view_as_complex(out)


def partsums_ex(torch, pix_re, pix_im=None, function="dft", orthogonal=False, K=None, N=None, partsum=(1, 1), offset=(0, 0), inverse=False):
    """The general form (dspfft_applybasis_partsums_ex): K = (Kw, Kh) basis functions, N = (Nw, Nh) blocks of `partsum` pixels;
    --inverse is K = image size, N = terms / partsum (applybasis.c:372-380) and moves --offset from the basis index to the block index
    (:416-420); pix_im: the imaginary part of a .coeff input."""
    lib = _lib.load()
    h, w, c = pix_re.shape
    assert c == 3 and pix_re.dtype == torch.float32 and pix_re.is_contiguous()
    func = FUNCTIONS.index(function)
    pw, ph = partsum
    kw, kh = K if K else (w, h)
    nw, nh = N if N else (w // pw, h // ph)
    out = torch.empty((kh, kw, nh, nw, 3, 2), dtype=torch.float32, device=pix_re.device)
    work = torch.empty(lib.dspfft_applybasis_work_floats_ex(w, h, kw, kh, nw, nh, func), dtype=torch.float32, device=pix_re.device)
    rc = lib.dspfft_applybasis_partsums_ex(out.data_ptr(), pix_re.data_ptr(), pix_im.data_ptr() if pix_im is not None else None, w, h, func, int(orthogonal),
                                           kw, kh, nw, nh, pw, ph, offset[0], offset[1], int(inverse), work.data_ptr(), None)
    if rc:
        raise DspfftError(lib.dspfft_zoom_last_error().decode())
    return torch.view_as_complex(out)


PLANES = ("real", "imaginary", "magnitude", "phase")
RESCALES = ("linear", "log", "gain", "level")
RANGES = ("shift", "abs", "invert", "hue")


def render(torch, parts, inverse=False, scale=1, padding=1, plane="real", rescale=("linear",), range_="shift", coeff_scale=1.0, insize_wh=1.0,
           padcolor=(0.0, 0.0, 0.0, 1.0)):
    """applybasis.c:392-442: RGBA float frame of the partial sums `parts` (complex [Kh, Kw, Nh, Nw, 3])"""
    lib = _lib.load()
    kh, kw, nh, nw, _ = parts.shape
    tw, th = (nw, nh) if inverse else (kw, kh)
    fw, fh = kw * nw * scale + padding * tw + padding, kh * nh * scale + padding * th + padding
    frame = torch.empty((fh, fw, 4), dtype=torch.float32, device=parts.device)
    pc = (C.c_float * 4)(*padcolor)
    r0 = RESCALES.index(rescale[0])
    r1 = RESCALES.index(rescale[1]) if len(rescale) > 1 else -1
    rc = lib.dspfft_applybasis_render(frame.data_ptr(), torch.view_as_real(parts).data_ptr(), kw, kh, nw, nh, int(inverse), scale, padding,
                                      PLANES.index(plane), r0, r1, RANGES.index(range_), float(coeff_scale), float(insize_wh), pc, None)
    if rc:
        raise DspfftError(lib.dspfft_zoom_last_error().decode())
    return frame


# ---- the `.coeff` file (applybasis.c:381-388 header, :443 body; read back at :319-338) ----
# coords {unsigned long long w, h} followed by w*h*3 `complex intermediate` values in the loop order of :410-413 (k_h, k_w, n_h, n_w, channel).
# `intermediate` depends on the build: applybasis/Makefile:1-2 builds INTERMEDIATE_PRECISION=L, so the files a default reference build
# writes and reads hold `complex long double` -- 32 bytes a value on x86-64 (two 80-bit x87 numbers in 16-byte slots).  That layout is the
# default here; "D" (16 bytes) and "F" (8) are the other builds' files.  A reader need not be told: the size of the file says which.
COEFF_DTYPES = {"L": "clongdouble", "D": "complex128", "F": "complex64"}


def _coeff_dtype(precision):
    import numpy as np
    dt = np.dtype(getattr(np, COEFF_DTYPES[precision]))
    if precision == "L" and dt.itemsize != 32:
        raise DspfftError("this platform's long double is not the 16-byte x87 format the reference's `.coeff` files hold")
    return dt


def write_coeff(path, parts_np, precision=None):
    """parts_np: complex [k_h][k_w][n_h][n_w][3] (the partial sums in loop order); precision: the INTERMEDIATE_PRECISION of the build that will read
    it.  None (default): "L", what the reference's default build reads (applybasis/Makefile:1-2) -- round 4 and before wrote "D"; on a platform whose
    long double is not the 16-byte x87 format (aarch64, MSVC) no "L" file can be written and the default falls back to "D" with a warning."""
    import warnings
    import numpy as np
    kh, kw, nh, nw, c = parts_np.shape
    assert c == 3
    if precision is None:
        precision = "L"
        if np.dtype(np.clongdouble).itemsize != 32:
            warnings.warn("long double is not the 16-byte x87 format here: writing a COEFF file for an INTERMEDIATE_PRECISION=D build of applybasis")
            precision = "D"
    with open(path, "wb") as f:
        np.array([nw * kw, nh * kh], dtype=np.uint64).tofile(f)          # dumpsize = {N.w K.w, N.h K.h}
        np.ascontiguousarray(parts_np).astype(_coeff_dtype(precision)).tofile(f)


def read_coeff(path, precision=None):
    """-> complex128 array [h][w][3] exactly as applybasis reads it back: `pixels[(y * insize.w + x) * 3 + j]`.  precision None: from the file's size"""
    import os
    import numpy as np
    with open(path, "rb") as f:
        w, h = (int(v) for v in np.fromfile(f, dtype=np.uint64, count=2))
        n = w * h * 3
        if n == 0:                                   # an empty frame: nothing to tell the precision by, nothing to read
            return np.zeros((h, w, 3), dtype=np.complex128)
        if precision is None:
            body = os.path.getsize(path) - 16
            precision = {32 * n: "L", 16 * n: "D", 8 * n: "F"}.get(body)
            if precision is None:
                raise ValueError(f".coeff file of {w} x {h} values with a body of {body} bytes: no build of the reference writes that")
        data = np.fromfile(f, dtype=_coeff_dtype(precision), count=n)
    if data.size != n:
        raise ValueError("short .coeff file")
    return data.astype(np.complex128).reshape(h, w, 3)
