"""Batches of small frames as ONE plan (guru-shaped batch over frames): roundtrip throughput.  python tools/bench_batches.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01

def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

for (h, w, c, F) in [(64, 64, 3, 4096), (100, 100, 3, 1000), (256, 256, 3, 256), (512, 512, 3, 64), (480, 640, 3, 64), (720, 1280, 3, 16), (1080, 1920, 3, 8), (2160, 3840, 3, 2)]:
    n = h * w * c
    x = torch.rand(F, h, w, c, device="cuda:0")
    dims = [(h, w * c, w * c), (w, c, c)]; how = [(c, 1, 1), (F, n, n)]
    f = Plan.guru(dims, how, [REDFT10] * 2); i = Plan.guru(dims, how, [REDFT01] * 2).set_scale(1.0 / (4.0 * h * w))
    ms = t(lambda: (f.execute(x.data_ptr()), i.execute(x.data_ptr())))
    print(json.dumps({"frames": F, "frame": f"{w}x{h}x{c}", "batch_MB": round(F * n * 4 / 1e6, 1), "roundtrip_us": round(ms * 1000, 1), "Mpix_s": round(F * h * w / ms / 1e3),
                      "frac_of_8TBps": round(F * n * 16 / ms / 1e6 / 8000, 3), "passes": [l.split()[2] for l in f.describe().splitlines()[1:]]}), flush=True)
