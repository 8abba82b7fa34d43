"""motion's block modes (--blocksize WxHxD) on a 1920x1080x256 luma volume: all axes of a block in one pass (block_core.h) against one
pass per axis (DSPFFT_NO_BLOCK=1); forward + inverse as two executes, and the whole pipeline (8-bit load -> REDFT10 -> quantiser ->
REDFT01 -> 8-bit store) as ONE pass.   python tools/bench_blocks.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01

def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

D, H, W = 256, 1080, 1920
x = torch.rand(D, H, W, device="cuda:0")
u8 = (torch.rand(D, H, W, device="cuda:0") * 255).to(torch.uint8)
o8 = torch.empty_like(u8)
n = D * H * W
for (bd, bh, bw) in [(8, 8, 8), (16, 8, 8), (4, 4, 4), (16, 16, 16), (1, 8, 8), (8, 12, 16), (256, 1, 1)]:
    dims = [(bd, H * W, H * W), (bh, W, W), (bw, 1, 1)]
    how = [(D // bd, bd * H * W, bd * H * W), (H // bh, bh * W, bh * W), (W // bw, bw, bw)]
    dims = [d for d in dims if d[0] > 1]
    line = {"block_dhw": [bd, bh, bw], "blocks": (D // bd) * (H // bh) * (W // bw)}
    for tag, env in (("fused", None), ("per_axis", "1")):
        if env: os.environ["DSPFFT_NO_BLOCK"] = env
        else: os.environ.pop("DSPFFT_NO_BLOCK", None)
        f = Plan.guru(dims, how, [REDFT10] * len(dims)); i = Plan.guru(dims, how, [REDFT01] * len(dims)).set_scale(1.0 / (2.0 ** len(dims) * bd * bh * bw))
        ms = t(lambda: (f.execute(x.data_ptr()), i.execute(x.data_ptr())))
        line[tag + "_roundtrip_ms"] = round(ms, 2)
        line[tag + "_roundtrip_GBps_at_16B"] = round(n * 16 / ms / 1e6, 1)
        if tag == "fused":
            line["passes"] = f.describe().splitlines()[1].split(" wgs=")[0]
            if "BLOCK" in f.describe():
                flt = dict(active=(bd, bh, bw), minbuf_hw=(bh, bw), block_depth=bd, band_begin=(0, 0, 0), band_end=(bd, bh, bw), quantizer=20.0)
                ms = t(lambda: f.roundtrip(i, x.data_ptr(), filter=flt))
                line["pipeline_f32_one_pass_ms"] = round(ms, 2)
                ms = t(lambda: f.roundtrip_u8(i, u8.data_ptr(), o8.data_ptr(), x.data_ptr(), 1.0, filter=flt))
                line["pipeline_u8_one_pass_ms"] = round(ms, 2)
                line["pipeline_u8_Gsamples_per_s"] = round(n / ms / 1e6, 1)
    print(json.dumps(line), flush=True)
