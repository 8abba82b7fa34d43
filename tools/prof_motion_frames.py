"""motion's default mode on 1080p luma, 256 frames in one batched plan: 8-bit in -> REDFT10 -> quantiser -> REDFT01 -> 8-bit out as
three launches (dspfft_execute_roundtrip_u8).  For rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01
d_, h, w = 256, 1080, 1920
v8 = (torch.rand(d_, h, w, device="cuda:0") * 255).to(torch.uint8); o8 = torch.empty_like(v8)
vol = torch.empty(d_, h, w, device="cuda:0")
coded = torch.zeros(1, dtype=torch.int64, device="cuda:0")
f2 = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=d_, idist=h * w, odist=h * w)
i2r = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=d_, idist=h * w, odist=h * w, first_axis_first=True).set_scale(1.0 / (4.0 * h * w))
flt2 = dict(active=(1, h, w), minbuf_hw=(h, w), block_depth=1, band_begin=(0, 0, 0), band_end=(1, h, w), quantizer=3.0)
print(f2.describe(), i2r.describe())
for _ in range(int(os.environ.get("REPS", "5"))):
    f2.roundtrip_u8(i2r, v8.data_ptr(), o8.data_ptr(), vol.data_ptr(), 1.0, filter=flt2, d_coded=coded.data_ptr())
torch.cuda.synchronize()
