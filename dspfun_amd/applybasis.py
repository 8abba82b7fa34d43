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
