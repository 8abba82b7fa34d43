"""ms per frame of BASELINE config 3's zoom (1920x1080x3 -> 7680x4320, scale 4, fast-transform path), HIP events over 200 frames on one stream."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd.zoom import Zoom
w, h = 1920, 1080
z = Zoom(torch, torch.rand(h, w, 3, device="cuda:0"))
for _ in range(20):
    z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), method="fft")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
e0.record()
for _ in range(n):
    z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), method="fft")
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / n:.4f} ms/frame")
