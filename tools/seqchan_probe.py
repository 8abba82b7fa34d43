"""VERDICT r2 item 5 as a measurement: "run the channels one after another through a single N/2-complex LDS plane".  The planar plan of
the same three planes IS that schedule with nothing added (one channel per line in LDS, a third of the LDS per workgroup, as many
workgroups per CU as then fit) and without what the interleaved variant would add (the whole interleaved line held in registers across
the three channel rounds).  If the planar passes do not beat the interleaved ones per byte, the variant cannot.
    python tools/seqchan_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd import Plan, REDFT10, REDFT01


def passes(plan, buf, reps=40):
    out = []
    for p in range(plan.num_passes):
        for _ in range(3):
            plan.execute_pass(p, buf.data_ptr())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            plan.execute_pass(p, buf.data_ptr())
        b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps * 1000)
    return out


for (h, w, dt, td) in ((4320, 7680, "f32", torch.float32), (2160, 3840, "f64", torch.float64), (2160, 3840, "f32", torch.float32)):
    for kind, name in ((REDFT10, "REDFT10"), (REDFT01, "REDFT01")):
        il = Plan.image(h, w, 3, kind, dtype=dt)
        pl = Plan.many_r2r([h, w], [kind] * 2, howmany=3, idist=h * w, odist=h * w, dtype=dt)
        x = torch.rand(h, w, 3, device="cuda:0", dtype=td) * 1e-3
        print(f"{w}x{h} {dt} {name}: interleaved passes {['%.1f' % t for t in passes(il, x)]} us   [{il.describe()}]")
        print(f"{w}x{h} {dt} {name}: planar x3     passes {['%.1f' % t for t in passes(pl, x)]} us   [{pl.describe()}]")
