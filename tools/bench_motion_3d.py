"""BASELINE config 5 read literally -- ONE 3-D block over the luma clip (1920 x 1080 x 256) on one GPU, motion.c:535-552,641-753 with -b = the clip:
ms per clip for (a) forward + inverse plans alone, (b) dspfft_execute_roundtrip with motion's quantiser (z pass fused), (c) the same from and to 8-bit frames.
    [DSPFFT_COL_KPREF=K DSPFFT_COL_TPREF=T] python tools/bench_motion_3d.py"""
import json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd import Plan, REDFT10, REDFT01
dev = torch.device("cuda", 0)
d_, h, w = 256, 1080, 1920
n = d_ * h * w
r2 = math.sqrt(2.0)


def timeit(fn, reps=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


vol = torch.rand(d_, h, w, device=dev) * 255
fwd = Plan.many_r2r([d_, h, w], [REDFT10] * 3).set_scale(2 * r2)
inv = Plan.many_r2r([d_, h, w], [REDFT01] * 3).set_scale(1.0 / (2 * r2) / (8.0 * d_ * h * w))
inv3 = Plan.many_r2r([d_, h, w], [REDFT01] * 3, first_axis_first=True).set_scale(1.0 / (2 * r2) / (8.0 * d_ * h * w))
for a in range(3):
    fwd.set_axis_scale0(a, 1.0, 1.0 / r2); inv.set_axis_scale0(a, r2, 1.0); inv3.set_axis_scale0(a, r2, 1.0)
flt = dict(active=(d_, h, w), minbuf_hw=(h, w), block_depth=d_, band_begin=(0, 0, 0), band_end=(d_, h, w), quantizer=3.0)
coded = torch.zeros(1, dtype=torch.int64, device=dev)
res = {"plan": fwd.describe().splitlines()[-1] if fwd.describe() else None}


def plain():
    fwd.execute(vol.data_ptr()); inv.execute(vol.data_ptr())


res["forward_inverse_ms"] = round(timeit(plain), 3)
vol.copy_(torch.rand(d_, h, w, device=dev) * 255)
res["roundtrip_quantised_ms"] = round(timeit(lambda: fwd.roundtrip(inv3, vol.data_ptr(), filter=flt, d_coded=coded.data_ptr())), 3)
v8 = (torch.rand(d_, h, w, device=dev) * 255).to(torch.uint8)
o8 = torch.empty_like(v8)
res["roundtrip_u8_ms"] = round(timeit(lambda: fwd.roundtrip_u8(inv3, v8.data_ptr(), o8.data_ptr(), vol.data_ptr(), 1.0, filter=flt, d_coded=coded.data_ptr())), 3)
res["max_abs_u8_change"] = int((o8.int() - v8.int()).abs().max())
res["u8_crc"] = int(o8.to(torch.int64).sum())
res["Msamples_per_s_u8"] = round(n / res["roundtrip_u8_ms"] / 1e3, 1)
print(json.dumps(res))
