"""Per-pass timings of the 2-D roundtrip at a given frame size (one frame, one stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
sizes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[1:] or ["4320x7680x3", "2160x3840x3", "1080x1920x3", "540x960x3", "1080x1920x1", "540x960x1"])]
for (h, w, c) in sizes:
    x = torch.rand(h, w, c, device="cuda:0")
    f = Plan.image(h, w, c, REDFT10); i = Plan.image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * h * w))
    tot = 0
    for plan, tag in ((f, "II "), (i, "III")):
        for k in range(plan.num_passes):
            us = t(lambda: plan.execute_pass(k, x.data_ptr())) * 1000
            tot += us
            print(f"{h}x{w}x{c} {tag} pass{k} {us:8.1f} us  {2*h*w*c*4/us/1e3:7.1f} GB/s moved   {plan.describe().splitlines()[1+k][:88]}")
    print(f"{h}x{w}x{c} roundtrip {tot:8.1f} us = {h*w/tot:8.1f} Mpix/s = {h*w*c*16/tot/1e3/8000:.3f} of 8 TB/s (algorithmic 16 B/sample)")
