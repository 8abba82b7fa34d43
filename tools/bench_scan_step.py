"""BASELINE config 4's fused frame step (7680x4320x3, zigzag, step 2^20, 32 frames): time per frame index with and without the
empty-tile skip (DSPFFT_NO_ZSKIP=1).  Under rocprofv3 --kernel-trace the two kernels of each step show up separately."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, _lib, REDFT10, REDFT01
L = _lib.load()
w, h, c = (int(v) for v in os.environ.get("SHAPE", "7680,4320,3").split(","))
step = 1 << 20 if w * h > (1 << 22) else (w * h + 31) // 32
nframes = (w * h + step - 1) // step
x = torch.rand(h, w, c, device="cuda:0")
coeffs = x.clone()
Plan.image(h, w, c, REDFT10).set_scale(1.0 / (4.0 * w * h)).execute(coeffs.data_ptr())
ids = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
assert L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, step, None) == 0
acc = torch.empty_like(coeffs); work = torch.empty_like(coeffs)
for mode in ("prepared", "skip", "dense"):
    if mode == "dense":
        os.environ["DSPFFT_NO_ZSKIP"] = "1"
    else:
        os.environ.pop("DSPFFT_NO_ZSKIP", None)
    inv = Plan.image(h, w, c, REDFT01)
    if mode == "prepared":
        inv.scan_prepare(ids.data_ptr(), c)
    per = []
    for rep in range(3):
        L.dspfft_broadcast_dc(acc.data_ptr(), coeffs.data_ptr(), w * h, c, None)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(nframes + 1)]
        ev[0].record()
        for f in range(nframes):
            inv.execute_masked_accumulate(coeffs.data_ptr(), work.data_ptr(), acc.data_ptr(), ids.data_ptr(), f, c)
            ev[f + 1].record()
        torch.cuda.synchronize()
        per = [ev[f].elapsed_time(ev[f + 1]) * 1e3 for f in range(nframes)]
    err = float((acc - x).abs().max())
    print(f"{mode}: mean {sum(per)/len(per):.0f} us/frame, err {err:.2e}; per frame (us): " + " ".join(f"{p:.0f}" for p in per), flush=True)
