import os, sys, math, torch
sys.path.insert(0, os.getcwd())
from dspfun_amd import Plan, REDFT10, REDFT01
h, w, nf = 1080, 1920, 256
fwd = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=nf, idist=h * w, odist=h * w).set_scale(2.0)
inv = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=nf, idist=h * w, odist=h * w, first_axis_first=True).set_scale(1.0 / 2.0 / (4.0 * h * w))
print(fwd.describe())
src = (torch.rand(nf, h, w, device="cuda:0") * 255).to(torch.uint8); dst = torch.empty_like(src); work = torch.empty(nf, h, w, device="cuda:0")
flt = dict(active=(1, h, w), minbuf_hw=(h, w), block_depth=1, band_begin=(0, 0, 0), band_end=(1, h, w), quantizer=20 * 8 * math.sqrt(w * h))
coded = torch.zeros(1, dtype=torch.int64, device="cuda:0")
def f(): fwd.roundtrip_u8(inv, src.data_ptr(), dst.data_ptr(), work.data_ptr(), 1.0, filter=flt, d_coded=coded.data_ptr())
for _ in range(3): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): f()
b.record(); torch.cuda.synchronize()
print("luma per-frame u8->u8: %.3f ms" % (a.elapsed_time(b) / 10))
