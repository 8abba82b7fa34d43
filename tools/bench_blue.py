"""Bluestein (O(N log N)) against the O(N^2) dense kernel on frame sizes with large prime factors.
python tools/bench_blue.py"""
import json, os, sys
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

for (h, w, c) in [(768, 1366, 3), (1087, 1933, 3), (1200, 1922, 3), (2161, 3841, 3), (683, 1031, 1)]:
    x = torch.rand(h, w, c, device="cuda:0")
    row = {"size": f"{w}x{h}x{c}"}
    for tag in ("bluestein", "dense"):
        if tag == "dense": os.environ["DSPFFT_NO_BLUESTEIN"] = "1"
        f = Plan.image(h, w, c, REDFT10); i = Plan.image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * h * w))
        os.environ.pop("DSPFFT_NO_BLUESTEIN", None)
        row[tag + "_us"] = round(t(lambda: (f.execute(x.data_ptr()), i.execute(x.data_ptr())), reps=5 if tag == "dense" else 10) * 1000, 1)
        if tag == "bluestein": row["plan"] = [l.split(" wgs")[0] for l in f.describe().splitlines()[1:]]
    row["speedup"] = round(row["dense_us"] / row["bluestein_us"], 1)
    print(json.dumps(row), flush=True)
