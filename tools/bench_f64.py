"""f64 (fftw_ API, spec/zoom default build) frame roundtrips: specialised double kernels vs the runtime-geometry ones (DSPFFT_NO_SPEC=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01

def run(H, W, C, env, two_streams=False):
    for k in ("DSPFFT_NO_SPEC",):
        os.environ.pop(k, None)
    os.environ.update(env)
    x = torch.rand(H, W, C, device="cuda:0", dtype=torch.float64)
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
    print(f"{H}x{W}x{C} {env or 'default'}: {best*1e6:.0f} us/roundtrip = {H*W*C*32/best/8e12*100:.1f} % of the {32*C} B/pixel roofline", flush=True)
    print("   ", fwd.describe().splitlines()[1:], flush=True)

if __name__ == "__main__":
    for (H, W, C) in ((2160, 3840, 3), (1080, 1920, 3), (2160, 3840, 1)):
        run(H, W, C, {})
        run(H, W, C, {"DSPFFT_NO_SPEC": "1"})
