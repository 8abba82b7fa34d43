import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd import Plan, REDFT10, REDFT01, _lib
L = _lib.load()
h, w = 4320, 7680
inv = Plan.many_r2r([h, w], [REDFT01] * 2)
print(inv.describe())
c = torch.rand(h, w, device="cuda:0"); work = torch.empty_like(c); acc = torch.zeros_like(c)
ids = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, 1 << 20, None)
def t(fn, reps=32):
    for k in range(3): fn(k)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(reps): fn(k)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000
f = lambda k: inv.execute_masked_accumulate(c.data_ptr(), work.data_ptr(), acc.data_ptr(), ids.data_ptr(), k % 32, 1)
print("planar fused step, no prepare: %.1f us" % t(f))
inv.scan_prepare(ids.data_ptr(), 1)
print("planar fused step, prepared:   %.1f us" % t(f))
print("plain inverse execute:          %.1f us" % t(lambda k: inv.execute(c.data_ptr())))
