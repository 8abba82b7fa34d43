"""f64 (fftw_ API, spec/zoom default build) 3840x2160x3 roundtrip on the runtime-geometry kernels under planner overrides."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01
H, W, C = 2160, 3840, 3
x = torch.rand(H, W, C, device="cuda:0", dtype=torch.float64)
def run(env):
    for k in ("DSPFFT_COL_K", "DSPFFT_COL_THREADS", "DSPFFT_ROW_THREADS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    fwd = Plan.image(H, W, C, REDFT10, dtype="f64")
    inv = Plan.image(H, W, C, REDFT01, dtype="f64").set_scale(1.0 / (4.0 * W * H))
    for _ in range(3):
        fwd.execute(x.data_ptr()); inv.execute(x.data_ptr())
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            fwd.execute(x.data_ptr()); inv.execute(x.data_ptr())
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    per = 'n/a'
    print(env, f"{best*1e6:.0f} us/roundtrip = {H*W*96/best/8e12*100:.1f} % of the 96 B/pixel roofline; passes {per}", flush=True)
    print("   ", fwd.describe().splitlines()[1:], flush=True)
run({})
for k in ("4", "6", "8"):
    for t in ("512", "1024"):
        run({"DSPFFT_COL_K": k, "DSPFFT_COL_THREADS": t})
for t in ("512", "1024"):
    run({"DSPFFT_COL_K": "4", "DSPFFT_COL_THREADS": "512", "DSPFFT_ROW_THREADS": t})
