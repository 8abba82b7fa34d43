import os, sys, torch
sys.path.insert(0, os.getcwd())
from dspfun_amd import Plan, REDFT10, REDFT01
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000
h, w, c = 4320, 7680, 3
x = torch.rand(h, w, c, device="cuda:0")
os.environ["DSPFFT_NO_SPLIT"] = "1"
f = Plan.image(h, w, c, REDFT10); i = Plan.image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * w * h))
print("8K f32 plain passes, DSPFFT_ROW_PERSIST=%s:" % os.environ.get("DSPFFT_ROW_PERSIST", "1"), [round(t(lambda p=p, k=k: p.execute_pass(k, x.data_ptr())), 1) for p in (f, i) for k in range(2)], "us per pass;", f.describe().splitlines()[1][:60])
