"""8K (7680x4320x3) and 4K roundtrip, one frame at a time on one stream: split (row pairs + half tiles) vs plain passes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01

def plans(h, w, c, nosplit):
    if nosplit:
        os.environ["DSPFFT_NO_SPLIT"] = "1"
    try:
        return Plan.image(h, w, c, REDFT10), Plan.image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * w * h))
    finally:
        os.environ.pop("DSPFFT_NO_SPLIT", None)

for (h, w) in ((4320, 7680), (2160, 3840)):
    x = torch.rand(h, w, 3, device="cuda:0")
    res = {}
    for name, ns in (("split", False), ("plain", True)):
        fwd, inv = plans(h, w, 3, ns)
        for _ in range(5):
            fwd.execute(x.data_ptr()); inv.execute(x.data_ptr())
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(30):
                fwd.execute(x.data_ptr()); inv.execute(x.data_ptr())
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 30)
        # per pass
        per = []
        for plan in (fwd, inv):
            for i in range(plan.num_passes):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3): plan.execute_pass(i, x.data_ptr())
                a.record()
                for _ in range(20): plan.execute_pass(i, x.data_ptr())
                b.record(); torch.cuda.synchronize()
                per.append(round(a.elapsed_time(b) / 20 * 1e3, 1))
        res[name] = (best, per)
        print(f"{w}x{h} {name}: {best*1e6:.1f} us per roundtrip = {h*w/best/1e6:.0f} Mpix/s; passes alone (us): {per}", flush=True)
