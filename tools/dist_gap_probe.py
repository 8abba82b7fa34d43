"""Probe (not product): where do ~1.4 ms go when bench.py's timed region runs in a process that has initialised RCCL?
Run plain (python tools/dist_gap_probe.py) and under torch.distributed.run with DIST=1."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd import Plan, REDFT10, REDFT01
from dspfun_amd.engine import Batch, Stream, Events
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
use_dist = os.environ.get("DIST") == "1"
if use_dist:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=dev)
H, W, C = 2160, 3840, 3
fwd = Plan.image(H, W, C, REDFT10)
inv = Plan.many_r2r([H, W], [REDFT01] * 2, howmany=C, istride=C, idist=1, ostride=C, odist=1, first_axis_first=True).set_scale(1.0 / (4.0 * W * H))
frames = torch.rand((4, H, W, C), device=dev)
st = [Stream(), Stream()]
batch = Batch([(pl, frames[f].data_ptr(), None, st[f % 2].handle) for f in range(4) for pl in (fwd, inv)])
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for mode in ("no barrier", "barrier before") if use_dist else ("no barrier",):
    for steps in (20, 20, 100):
        batch.run_repeat(60, 8)
        torch.cuda.synchronize()
        if mode == "barrier before":
            dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        batch.run_repeat(steps, 8)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"dist={use_dist} {mode}: {steps} steps: enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms = {steps*4*H*W/1e6/(t2-t0):.0f} Mpix/s", flush=True)
if use_dist:
    dist.destroy_process_group()
