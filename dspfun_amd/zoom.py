"""Host-side mirror of zoom's frame computation (zoom/zoom.c:263-265 forward transform,
:347-375 basis generation and dense separable product) over device memory."""
import ctypes as C

from . import _lib
from .engine import Plan, DspfftError, REDFT10

INTERPOLATED, CENTERED, NATIVE = 0, 1, 2     # zoom/zoom.c:20-26
CACHE_PLANS = 4       # frame plans kept per path: an animated zoom (a new scale every frame) would otherwise keep a plan and a frame-sized work
                      # buffer (about 100 MB at 1080p -> 4K) alive per frame; the least recently used one is destroyed


class _PlanCache:
    """key -> (handle, work) or None ("does not apply"), at most CACHE_PLANS live handles, least recently used destroyed first"""

    def __init__(self, destroy):
        self.destroy, self.d = destroy, {}

    def get(self, key, make):
        if key in self.d:
            self.d[key] = self.d.pop(key)                  # most recently used last
            return self.d[key]
        v = self.d[key] = make()
        live = [k for k, e in self.d.items() if e is not None]
        for k in live[:max(0, len(live) - CACHE_PLANS)]:
            self.destroy(self.d.pop(k)[0])
        for k in [k for k, e in self.d.items() if e is None][:-64]:      # (the "does not apply" marks hold nothing; bounded all the same)
            del self.d[k]
        return v

    def __len__(self):
        return len(self.d)

    def values(self):
        return self.d.values()

    def clear(self):
        for e in self.d.values():
            if e is not None:
                self.destroy(e[0])
        self.d = {}


class Zoom:
    """coeffs = REDFT10^2(image) once (zoom.c:263-265); frame(...) per output frame (zoom.c:320-375)."""

    def __init__(self, torch, image_hwc):
        self.torch = torch
        self.lib = _lib.load()
        self.h, self.w, c = image_hwc.shape
        assert c == 3 and image_hwc.dtype == torch.float32 and image_hwc.is_cuda
        self.coeffs = image_hwc.contiguous().clone()
        Plan.image(self.h, self.w, 3, REDFT10).execute(self.coeffs.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)

    def _frame_fft(self, vw, vh, xscale, yscale, vx, vy, basis_type):
        key = (vw, vh, tuple(xscale), tuple(yscale), basis_type)
        if not hasattr(self, "_fft"):
            self._fft = _PlanCache(self.lib.dspfft_zoomfft_destroy)

        def make():
            z = C.c_void_p()
            rc = self.lib.dspfft_zoomfft_create(C.byref(z), self.w, self.h, basis_type, xscale[0], xscale[1], yscale[0], yscale[1], vw, vh)
            if rc == -2:
                return None                        # this scale / basis / viewport keeps the dense product
            if rc:
                raise DspfftError(self.lib.dspfft_zoomfft_last_error().decode())
            return (z, self.torch.empty(self.lib.dspfft_zoomfft_work_floats(z), dtype=self.torch.float32, device=self.coeffs.device))
        e = self._fft.get(key, make)
        if e is None:
            return None
        z, work = e
        out = self.torch.empty((vh, vw, 3), dtype=self.torch.float32, device=self.coeffs.device)
        if self.lib.dspfft_zoomfft_execute(z, self.coeffs.data_ptr(), float(vx), float(vy), out.data_ptr(), work.data_ptr(),
                                           self.torch.cuda.current_stream().cuda_stream):
            raise DspfftError(self.lib.dspfft_zoomfft_last_error().decode())
        return out

    def _frame_czt(self, vw, vh, xscale, yscale, vx, vy, basis_type):
        """chirp-z transforms along both axes (dspfft_zoomczt_*): any scale, offset and basis; None when an axis is too long for the
        listed convolution lengths"""
        key = (vw, vh, tuple(xscale), tuple(yscale), basis_type)
        if not hasattr(self, "_czt"):
            self._czt = _PlanCache(self.lib.dspfft_zoomczt_destroy)

        def make():
            z = C.c_void_p()
            rc = self.lib.dspfft_zoomczt_create(C.byref(z), self.w, self.h, basis_type, xscale[0], xscale[1], yscale[0], yscale[1], vw, vh)
            if rc == -2:
                return None
            if rc:
                raise DspfftError(self.lib.dspfft_zoomfft_last_error().decode())
            return (z, self.torch.empty(self.lib.dspfft_zoomczt_work_floats(z), dtype=self.torch.float32, device=self.coeffs.device))
        e = self._czt.get(key, make)
        if e is None:
            return None
        z, work = e
        out = self.torch.empty((vh, vw, 3), dtype=self.torch.float32, device=self.coeffs.device)
        if self.lib.dspfft_zoomczt_execute(z, self.coeffs.data_ptr(), float(vx), float(vy), out.data_ptr(), work.data_ptr(),
                                           self.torch.cuda.current_stream().cuda_stream):
            raise DspfftError(self.lib.dspfft_zoomfft_last_error().decode())
        return out

    def __del__(self):
        try:
            for c in (getattr(self, "_fft", None), getattr(self, "_czt", None)):
                if c is not None:
                    c.clear()
        except Exception:
            pass

    def _basis(self, basis_type, num, den, offset, nvectors, length):
        nc = self.lib.dspfft_zoom_ncomponents(num, den, length)
        b = self.torch.empty(nvectors * nc, dtype=self.torch.float32, device=self.coeffs.device)
        if self.lib.dspfft_zoom_basis(b.data_ptr(), basis_type, num, den, offset, nvectors, length, None):
            raise DspfftError(self.lib.dspfft_zoom_last_error().decode())
        return b, nc

    def frame(self, vw, vh, xscale=(1.0, 1.0), yscale=(1.0, 1.0), vx=0.0, vy=0.0, basis_type=INTERPOLATED, method="auto"):
        """one output frame: (vh, vw, 3) f32.  method "auto": fast transforms on the DCT-III grid (dspfft_zoomfft_*) when the scaled lengths
        are integers and the basis is interpolated or native; chirp-z transforms (dspfft_zoomczt_*) for every other scale and the centered
        basis; the dense MFMA product only for axes beyond the listed convolution lengths.  "fft" / "czt" / "gemm" force one (fft and czt
        raise if they do not apply)."""
        torch = self.torch
        if method in ("auto", "fft"):
            out = self._frame_fft(vw, vh, xscale, yscale, vx, vy, basis_type)
            if out is not None:
                return out
            if method == "fft":
                raise DspfftError(self.lib.dspfft_zoomfft_last_error().decode())
        if method in ("auto", "czt"):
            out = self._frame_czt(vw, vh, xscale, yscale, vx, vy, basis_type)
            if out is not None:
                return out
            if method == "czt":
                raise DspfftError(self.lib.dspfft_zoomfft_last_error().decode())
        xb, cw = self._basis(basis_type, xscale[0], xscale[1], vx, vw, self.w)
        yb, ch = self._basis(basis_type, yscale[0], yscale[1], vy, vh, self.h)
        out = torch.empty((vh, vw, 3), dtype=torch.float32, device=self.coeffs.device)
        work = torch.empty(self.lib.dspfft_zoom_work_floats(self.w, self.h, ch, vw), dtype=torch.float32, device=self.coeffs.device)
        rc = self.lib.dspfft_zoom_product(self.coeffs.data_ptr(), self.w, self.h, xb.data_ptr(), cw, yb.data_ptr(), ch,
                                          out.data_ptr(), vw, vh, work.data_ptr(), None)
        if rc:
            raise DspfftError(self.lib.dspfft_zoom_last_error().decode())
        return out
