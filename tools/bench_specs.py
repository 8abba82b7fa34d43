"""Roundtrip time of the common frame sizes on the specialised kernels and (DSPFFT_NO_SPEC=1 at plan time) on the
runtime-geometry kernels.  One frame per launch, one stream.  python tools/bench_specs.py"""
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

sizes = [(256, 256, 3), (480, 640, 3), (480, 720, 3), (512, 512, 3), (540, 960, 3), (720, 1280, 3), (1024, 1024, 3), (1080, 1920, 3), (1440, 2560, 3),
         (2048, 2048, 3), (2160, 3840, 3), (2160, 4096, 3), (4096, 4096, 3), (4320, 7680, 3), (720, 1280, 1), (1080, 1920, 1),
         (600, 800, 3), (768, 1024, 3), (900, 1440, 3), (900, 1600, 3), (960, 1280, 3), (1200, 1600, 3), (1200, 1920, 3), (1600, 2560, 3), (1800, 3200, 3),
         (2880, 5120, 3), (2160, 3840, 1), (1440, 2560, 1), (2160, 4096, 1), (1080, 2048, 1), (768, 1024, 1)]
if os.environ.get("SIZES") == "new":
    sizes = sizes[16:]
out = []
for (h, w, c) in sizes:
    x = torch.rand(h, w, c, device="cuda:0")
    row = {"size": f"{w}x{h}x{c}"}
    for tag, nospec in (("spec", False), ("generic", True)):
        if nospec: os.environ["DSPFFT_NO_SPEC"] = "1"
        f = Plan.image(h, w, c, REDFT10); i = Plan.image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * h * w))
        os.environ.pop("DSPFFT_NO_SPEC", None)
        ms = t(lambda: (f.execute(x.data_ptr()), i.execute(x.data_ptr())))
        row[tag + "_us"] = round(ms * 1000, 1)
        if not nospec: row["kernels"] = f.describe().count("*")
    row["speedup"] = round(row["generic_us"] / row["spec_us"], 2)
    row["Mpix_s"] = round(h * w / row["spec_us"], 1)
    row["frac_of_8TBps"] = round(h * w * c * 16 / row["spec_us"] / 1e3 / 8000, 3)
    out.append(row)
    print(json.dumps(row), flush=True)
