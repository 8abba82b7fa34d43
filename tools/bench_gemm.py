import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dspfun_amd import _lib
from dspfun_amd.zoom import Zoom
if os.environ.get("BENCH_LIB"):          # same-box A/B against another build of the library (tools/ab_oldlib.sh)
    import ctypes
    _lib._lib = _lib.bind(ctypes.CDLL(os.environ["BENCH_LIB"]))
L = _lib.load()
def t(fn, reps=40):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
w, h = 1920, 1080
z = Zoom(torch, torch.rand(h, w, 3, device="cuda:0"))
vw, vh = 4 * w, 4 * h
xb, cw = z._basis(0, 4.0, 1.0, 0.0, vw, w); yb, ch = z._basis(0, 4.0, 1.0, 0.0, vh, h)
print("basis gen ms", t(lambda: (z._basis(0, 4.0, 1.0, 0.0, vw, w), z._basis(0, 4.0, 1.0, 0.0, vh, h))))
out = torch.empty((vh, vw, 3), device="cuda:0"); work = torch.empty(L.dspfft_zoom_work_floats(w, h, ch, vw), device="cuda:0")
ms = t(lambda: L.dspfft_zoom_product(z.coeffs.data_ptr(), w, h, xb.data_ptr(), cw, yb.data_ptr(), ch, out.data_ptr(), vw, vh, work.data_ptr(), None))
fl = 2.0 * 3 * (ch * vw * cw + vh * vw * ch)
print("product ms", ms, "TF", fl / ms / 1e9)
# plain GEMMs
A = torch.rand(7680, 1920, device="cuda:0"); B = torch.rand(1080, 1920, device="cuda:0"); Cc = torch.empty(7680, 1080, device="cuda:0")
ms = t(lambda: L.dspfft_gemm_nt_f32(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), 7680, 1080, 1920, 1920, 1920, 1080, 1, 1, 0, 0, 0, 1.0, None))
print("gemm1 7680x1080x1920 ms", ms, "TF", 2 * 7680 * 1080 * 1920 / ms / 1e9)
A2 = torch.rand(4320, 1080, device="cuda:0"); B2 = torch.rand(7680, 1080, device="cuda:0"); C2 = torch.empty(4320, 7680 * 3, device="cuda:0")
ms = t(lambda: L.dspfft_gemm_nt_f32(A2.data_ptr(), B2.data_ptr(), C2.data_ptr(), 4320, 7680, 1080, 1080, 1080, 7680 * 3, 3, 1, 0, 0, 0, 1.0, None))
print("gemm2 4320x7680x1080 cs=3 ms", ms, "TF", 2 * 4320 * 7680 * 1080 / ms / 1e9)
ms = t(lambda: L.dspfft_gemm_nt_f32(A2.data_ptr(), B2.data_ptr(), C2.data_ptr(), 4320, 7680, 1080, 1080, 1080, 7680, 1, 1, 0, 0, 0, 1.0, None))
print("gemm2 4320x7680x1080 cs=1 ms", ms, "TF", 2 * 4320 * 7680 * 1080 / ms / 1e9)
A3 = torch.rand(8192, 4096, device="cuda:0"); B3 = torch.rand(8192, 4096, device="cuda:0"); C3 = torch.empty(8192, 8192, device="cuda:0")
ms = t(lambda: L.dspfft_gemm_nt_f32(A3.data_ptr(), B3.data_ptr(), C3.data_ptr(), 8192, 8192, 4096, 4096, 4096, 8192, 1, 1, 0, 0, 0, 1.0, None))
print("gemm 8192x8192x4096 ms", ms, "TF", 2 * 8192 * 8192 * 4096 / ms / 1e9)

# the two products of a config-3 frame the way round 3 launched them: three channels per launch, the second storing with cs = 3
planes = torch.rand(3, h, w, device="cuda:0"); Tt = torch.empty(3, vw, ch, device="cuda:0")
ms = t(lambda: L.dspfft_gemm_nt_f32(xb.data_ptr(), planes.data_ptr(), Tt.data_ptr(), vw, ch, cw, cw, w, ch, 1, 3, 0, w * h, vw * ch, 1.0, None))
print("product 1, one launch of 3 channels (7680x1080x1920 x 3) ms", ms, "TF", 3 * 2 * vw * ch * cw / ms / 1e9)
ms = t(lambda: L.dspfft_gemm_nt_f32(yb.data_ptr(), Tt.data_ptr(), out.data_ptr(), vh, vw, ch, ch, ch, vw * 3, 3, 3, 0, vw * ch, 1, 1.0, None))
print("product 2, 3 channels with the interleaved store (4320x7680x1080 x 3, cs=3) ms", ms, "TF", 3 * 2 * vh * vw * ch / ms / 1e9)
# ... and as dspfft_zoom_product launches them now: Tt's rows are (x, channel), so both are ONE plain product with contiguous stores
ms = t(lambda: L.dspfft_gemm_nt_f32(xb.data_ptr(), planes.data_ptr(), Tt.data_ptr(), vw, 3 * ch, cw, cw, w, 3 * ch, 1, 1, 0, 0, 0, 1.0, None))
print("product 1 as one matrix (7680x3240x1920) ms", ms, "TF", 3 * 2 * vw * ch * cw / ms / 1e9)
ms = t(lambda: L.dspfft_gemm_nt_f32(yb.data_ptr(), Tt.data_ptr(), out.data_ptr(), vh, 3 * vw, ch, ch, ch, vw * 3, 1, 1, 0, 0, 0, 1.0, None))
print("product 2 as one matrix (4320x23040x1080) ms", ms, "TF", 3 * 2 * vh * vw * ch / ms / 1e9)
A4 = torch.rand(4096, 4096, device="cuda:0"); B4 = torch.rand(4096, 4096, device="cuda:0"); C4 = torch.empty(4096, 4096, device="cuda:0")
ms = t(lambda: L.dspfft_gemm_nt_f32(A4.data_ptr(), B4.data_ptr(), C4.data_ptr(), 4096, 4096, 4096, 4096, 4096, 4096, 1, 1, 0, 0, 0, 1.0, None))
print("gemm 4096^3 ms", ms, "TF", 2 * 4096 ** 3 / ms / 1e9)

# applybasis: the full dct2 spectrum of an n x n image (two batched products of n^3 per channel)
from dspfun_amd.applybasis import partsums
for n in (512, 1024, 2048):
    img = torch.rand(n, n, 3, device="cuda:0") * 2 - 1
    ms = t(lambda: partsums(torch, img, "dct2", True, None, (n, n)), reps=10)
    print("applybasis dct2 spectrum of %d^2 ms" % n, ms, "TF", 3 * 2 * 2 * n ** 3 / ms / 1e9)
