"""motion's block modes as single plans (guru-shaped batch): all WxHxD blocks of a 1920x1080x256 luma volume, roundtrip.
python tools/bench_blocks.py"""
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
n = D * H * W
for (bd, bh, bw) in [(8, 8, 8), (16, 8, 8), (4, 4, 4), (8, 12, 16), (256, 1, 1)]:
    dims = [(bd, H * W, H * W), (bh, W, W), (bw, 1, 1)]
    how = [(D // bd, bd * H * W, bd * H * W), (H // bh, bh * W, bh * W), (W // bw, bw, bw)]
    dims = [d for d in dims if d[0] > 1]
    f = Plan.guru(dims, how, [REDFT10] * len(dims)); i = Plan.guru(dims, how, [REDFT01] * len(dims))
    ms = t(lambda: (f.execute(x.data_ptr()), i.execute(x.data_ptr())))
    print(json.dumps({"block_dhw": [bd, bh, bw], "blocks": (D // bd) * (H // bh) * (W // bw), "roundtrip_ms": round(ms, 2), "Gsamples_per_s": round(n / ms / 1e6, 1),
                      "algorithmic_GBps": round(n * 16 / ms / 1e6, 1), "passes": [l.split(" lines")[0] for l in f.describe().splitlines()[1:]]}), flush=True)
