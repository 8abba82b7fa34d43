"""Runs the MFMA paths (zoom C3 product, applybasis full-spectrum DCT2 of 512^2, 1024^2 and 2048^2 images, a plain 8192x8192x4096 GEMM)
a few times for rocprofv3 --pmc (MFMA utilisation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import _lib
from dspfun_amd.zoom import Zoom
from dspfun_amd.applybasis import partsums
L = _lib.load()
w, h = 1920, 1080
z = Zoom(torch, torch.rand(h, w, 3, device="cuda:0"))
for _ in range(3):
    z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), method="gemm")
for n in (512, 1024, 2048):          # applybasis: the full dct2 spectrum of an n x n image (two n^3 products per channel)
    img = torch.rand(n, n, 3, device="cuda:0") * 2 - 1
    for _ in range(3):
        partsums(torch, img, "dct2", True, None, (n, n))
    del img
A = torch.rand(8192, 4096, device="cuda:0"); B = torch.rand(8192, 4096, device="cuda:0"); C = torch.empty(8192, 8192, device="cuda:0")
for _ in range(3):
    L.dspfft_gemm_nt_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), 8192, 8192, 4096, 4096, 4096, 8192, 1, 1, 0, 0, 0, 1.0, None)
torch.cuda.synchronize()
