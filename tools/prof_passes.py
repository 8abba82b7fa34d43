"""Runs each axis pass of the 4K roundtrip a few times (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01
H, W, C = (int(v) for v in os.environ.get("SHAPE", "2160,3840,3").split(","))
x = torch.rand(H, W, C, device="cuda:0")
fwd = Plan.image(H, W, C, REDFT10)
inv = Plan.image(H, W, C, REDFT01).set_scale(1.0 / (4.0 * W * H))
print(fwd.describe())
for _ in range(int(os.environ.get("REPS", "6"))):
    fwd.execute(x.data_ptr())
    inv.execute(x.data_ptr())
torch.cuda.synchronize()
