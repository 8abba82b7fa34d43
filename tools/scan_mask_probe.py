"""Where the fused scan step's masked column pass spends its time (8K RGB, zigzag, step 2^20), per mode (argv[1]):
   normal   frames 0..31 in turn, tile ranges prepared (what host/scan_dev.c and ChannelShardedScan run)
   idsonly  a frame id nobody owns, NO tile ranges: every workgroup reads its owner ids, finds nothing and leaves (no transform, no store)
   late     frames 16..31 only (every column tile is touched)
Run under rocprofv3 --kernel-trace --stats for the per-kernel split; prints the step's event time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd import Plan, REDFT10, REDFT01, _lib
L = _lib.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "normal"
h, w, c = 4320, 7680, 3
inv = Plan.image(h, w, c, REDFT01)
co = torch.rand(h, w, c, device="cuda:0"); work = torch.empty_like(co); acc = torch.zeros_like(co)
ids = torch.zeros(w * h, dtype=torch.int32, device="cuda:0")
L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, 1 << 20, None)
if mode != "idsonly":
    inv.scan_prepare(ids.data_ptr(), c)
frame = {"normal": lambda k: k % 32, "idsonly": lambda k: 99, "late": lambda k: 16 + k % 16}[mode]
f = lambda k: inv.execute_masked_accumulate(co.data_ptr(), work.data_ptr(), acc.data_ptr(), ids.data_ptr(), frame(k), c)
for k in range(4): f(k)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for k in range(64): f(k)
b.record(); torch.cuda.synchronize()
print(mode, "step: %.1f us" % (a.elapsed_time(b) / 64 * 1000))
