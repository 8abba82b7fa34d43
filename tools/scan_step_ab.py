"""The fused scan step of BASELINE config 4 (7680x4320 RGB, zigzag, 2^20 coefficients per frame) alone: us per step over the 32 frames, REPS times.
    DSPFFT_PAIR_PIPE_SCAN=0|1 python tools/scan_step_ab.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd import Plan, REDFT10, REDFT01, _lib
L = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
planar = len(sys.argv) > 2 and sys.argv[2] == "planar"          # the three colour planes one by one, as a rank of dist.ChannelShardedScan holds them on N > 1 GPUs
h, w, c = 4320, 7680, 3
ids = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, 1 << 20, None)
if planar:
    x = torch.rand(c, h, w, device="cuda:0")
    inv = Plan.many_r2r([h, w], [REDFT01] * 2); inv.scan_prepare(ids.data_ptr(), 1)
    acc = torch.zeros_like(x); work = torch.empty_like(x[0])
    def frames():
        for k in range(32):
            for z in range(c):
                inv.execute_masked_accumulate(x[z].data_ptr(), work.data_ptr(), acc[z].data_ptr(), ids.data_ptr(), k, 1)
else:
    x = torch.rand(h, w, c, device="cuda:0")
    inv = Plan.image(h, w, c, REDFT01); inv.scan_prepare(ids.data_ptr(), c)
    acc = torch.zeros_like(x); work = torch.empty_like(x)
    def frames():
        for k in range(32):
            inv.execute_masked_accumulate(x.data_ptr(), work.data_ptr(), acc.data_ptr(), ids.data_ptr(), k, c)
frames(); frames()
torch.cuda.synchronize()
out = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); frames(); b.record(); torch.cuda.synchronize()
    out.append(a.elapsed_time(b) / 32 * 1000)
print(("planar planes: " if planar else "") + "fused scan step, us per frame over 32 frames:", " ".join(f"{v:.1f}" for v in out), " min %.1f" % min(out))
