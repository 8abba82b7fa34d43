"""BASELINE config 3 (1920x1080 RGB -> 7680x4320, scale 4) by fast transforms, steady state: ms per frame over >= 40 ms of frames.
   python tools/zoom_frame_bench.py            # the duo row kernel (dspfft_cosrows_*) for the x stage
   DSPFFT_ZOOM_XROWS=0 python tools/zoom_frame_bench.py   # round 3's two-transform row pass (dspfft_execute_sum2)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd.zoom import Zoom
w, h = 1920, 1080
z = Zoom(torch, torch.rand(h, w, 3, device="cuda:0"))
out = {}
for name, args in (("whole frame 7680x4320", (4 * w, 4 * h, 0.0, 0.0)), ("panned 7680x4320 viewport at (100.25, 50.5)... clipped to 4000x3000", (4000, 3000, 100.25, 50.5))):
    f = lambda: z.frame(args[0], args[1], (4.0, 1.0), (4.0, 1.0), vx=args[2], vy=args[3], method="fft")
    for _ in range(300):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        f()
    b.record(); torch.cuda.synchronize()
    out[name] = round(a.elapsed_time(b) / 200, 4)
print(json.dumps({"x_stage": "two-transform row pass (sum2)" if os.environ.get("DSPFFT_ZOOM_XROWS") == "0" else "duo row kernel (cosrows)", "ms_per_frame": out}))
