import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd.zoom import Zoom
w, h = 1920, 1080
z = Zoom(torch, torch.rand(h, w, 3, device="cuda:0"))
def t(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for (vw, vh) in ((7680, 4320), (1920, 1080), (960, 540), (480, 270)):
    f = t(lambda: z.frame(vw, vh, (4.0, 1.0), (4.0, 1.0), 100.0, 50.0, method="fft"))
    g = t(lambda: z.frame(vw, vh, (4.0, 1.0), (4.0, 1.0), 100.0, 50.0, method="gemm"), reps=20)
    print(f"viewport {vw}x{vh} of the 7680x4320 scaled image: fast transforms {f:.3f} ms, dense product {g:.3f} ms")
