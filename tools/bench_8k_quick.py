"""8K f32 roundtrip (split passes) and the fused scan step, per-pass times: a quick A/B harness for row-pair / half-tile changes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd import Plan, REDFT10, REDFT01, _lib
L = _lib.load()
def t(fn, reps=100):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000
h, w, c = 4320, 7680, 3
x = torch.rand(h, w, c, device="cuda:0")
f = Plan.image(h, w, c, REDFT10); i = Plan.image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * w * h))
print(f.describe().splitlines()[1][:90])
print("8K roundtrip: %.1f us" % t(lambda: (f.execute(x.data_ptr()), i.execute(x.data_ptr()))), " passes:", [round(t(lambda p=p, k=k: p.execute_pass(k, x.data_ptr())), 1) for p in (f, i) for k in range(2)])
ids = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, 1 << 20, None)
inv = Plan.image(h, w, c, REDFT01); inv.scan_prepare(ids.data_ptr(), c)
acc = torch.zeros_like(x); work = torch.empty_like(x)
k = [0]
def step():
    inv.execute_masked_accumulate(x.data_ptr(), work.data_ptr(), acc.data_ptr(), ids.data_ptr(), k[0] % 32, c); k[0] += 1
print("fused scan step: %.1f us" % t(step, 32))
