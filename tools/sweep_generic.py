"""Sweep the generic kernels' launch choices (threads, Bg, K) through the DSPFFT_* plan-time overrides."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01
os.environ["DSPFFT_NO_SPEC"] = "1"
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000
sizes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]] or [(2160, 3840, 3), (1000, 1500, 3), (1536, 2048, 3), (3000, 4000, 3), (600, 800, 3)]
for (h, w, c) in sizes:
    x = torch.rand(h, w, c, device="cuda:0")
    for thr in (256, 512, 1024):
        os.environ["DSPFFT_ROW_THREADS"] = str(thr)
        f = Plan.image(h, w, c, REDFT10); i = Plan.image(h, w, c, REDFT01)
        print(f"{h}x{w}x{c} ROW thr={thr}: II {t(lambda: f.execute_pass(0, x.data_ptr())):7.1f} III {t(lambda: i.execute_pass(0, x.data_ptr())):7.1f} us   {f.describe().splitlines()[1][:70]}", flush=True)
    os.environ.pop("DSPFFT_ROW_THREADS")
    for K, thr in itertools.product((4, 8, 16), (256, 512, 1024)):
        os.environ["DSPFFT_COL_K"] = str(K); os.environ["DSPFFT_COL_THREADS"] = str(thr)
        f = Plan.image(h, w, c, REDFT10); i = Plan.image(h, w, c, REDFT01)
        print(f"{h}x{w}x{c} COL K={K} thr={thr}: II {t(lambda: f.execute_pass(1, x.data_ptr())):7.1f} III {t(lambda: i.execute_pass(1, x.data_ptr())):7.1f} us   {f.describe().splitlines()[2][:80]}", flush=True)
    os.environ.pop("DSPFFT_COL_K"); os.environ.pop("DSPFFT_COL_THREADS")
