import os, sys, json
sys.path.insert(0, os.getcwd())
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
for (h, w) in ((4320, 7680), (1080, 1920)):
    x = torch.rand(8 if h == 1080 else 1, h, w, 3, device="cuda:0")
    f = Plan.image(h, w, 3, REDFT10); i = Plan.image(h, w, 3, REDFT01).set_scale(1.0 / (4.0 * h * w))
    def rt():
        for k in range(x.shape[0]):
            f.execute(x[k].data_ptr()); i.execute(x[k].data_ptr())
    ms = t(rt) / x.shape[0]
    print(h, w, "split" if os.environ.get("DSPFFT_NO_SPLIT") != "1" else "nosplit", round(ms * 1000, 1), "us/roundtrip", round(h * w / ms / 1e3, 1), "Mpix/s", round(h * w * 48 / ms / 1e3 / 8e6, 4), "of roofline")
    print(f.describe().splitlines()[2][:100])
