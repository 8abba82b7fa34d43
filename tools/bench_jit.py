"""Frame sizes without an entry in spec_list.h: runtime-geometry kernels against kernels compiled at plan time (DSPFFT_JIT=1, hiprtc)."""
import json, os, sys, time
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

tune = os.environ.get("TUNE") == "1"
for (h, w, c, dt) in [(1000, 1500, 3, "f32"), (1500, 2000, 3, "f32"), (2000, 3000, 3, "f32"), (1350, 2400, 3, "f32"), (1500, 2000, 1, "f32"), (3000, 4000, 3, "f32"), (1500, 2000, 3, "f64")]:
    x = torch.rand(h, w, c, device="cuda:0", dtype=torch.float64 if dt == "f64" else torch.float32)
    row = {"size": f"{w}x{h}x{c} {dt}"}
    for tag, jit in (("generic", "0"), ("jit", "1")) + ((("tuned", "1"),) if tune else ()):
        os.environ["DSPFFT_JIT"] = jit
        os.environ["DSPFFT_JIT_TUNE"] = "1" if tag == "tuned" else "2"
        t0 = time.perf_counter()
        f = Plan.image(h, w, c, REDFT10, dtype=dt); i = Plan.image(h, w, c, REDFT01, dtype=dt).set_scale(1.0 / (4.0 * h * w))
        row[tag + "_plan_s"] = round(time.perf_counter() - t0, 2)
        row[tag + "_us"] = round(t(lambda: (f.execute(x.data_ptr()), i.execute(x.data_ptr()))) * 1000, 1)
        if jit == "1": row["plan" if tag == "jit" else "tuned_plan"] = [l.split(" compiled at plan time: ")[-1].split(",  ")[0][:70] for l in f.describe().splitlines()[1:]]
    row["speedup"] = round(row["generic_us"] / row["jit_us"], 2)
    print(json.dumps(row), flush=True)
