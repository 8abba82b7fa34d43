"""Probe (not product): why does the same dspfft_execute_many_repeat call enqueue ~6x slower from bench.py than from tools/cbench.c?
Varies one thing at a time: torch streams vs raw HIP streams, torch memory vs hipMalloc, events on/off."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01
from dspfun_amd.engine import Batch

H, W, Cc = 2160, 3840, 3
hip = C.CDLL("libamdhip64.so")
fwd = Plan.image(H, W, Cc, REDFT10)
inv = Plan.many_r2r([H, W], [REDFT01] * 2, howmany=Cc, istride=Cc, idist=1, ostride=Cc, odist=1, first_axis_first=True).set_scale(1.0 / (4.0 * W * H))
frames = torch.rand((4, H, W, Cc), device="cuda:0")
torch.cuda.synchronize()


def raw_stream():
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    return s.value


def run(label, handles, steps=300):
    batch = Batch([(pl, frames[f].data_ptr(), None, handles[f % 2]) for f in range(4) for pl in (fwd, inv)])
    batch.run_repeat(60, 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch.run_repeat(steps, 8)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label}: {steps * 4 * H * W / 1e6 / (t2 - t0):.0f} Mpix/s, enqueue {(t1 - t0) / steps * 1e3:.4f} ms/step", flush=True)


side = [torch.cuda.Stream() for _ in range(2)]
raw = [raw_stream(), raw_stream()]
for rnd in range(2):
    run("torch streams", [s.cuda_stream for s in side])
    run("raw HIP streams", raw)
