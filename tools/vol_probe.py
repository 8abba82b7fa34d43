import os, sys, json, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from bench_motion import volume_bench, frames_bench
dev = torch.device("cuda", 0)
print("fresh process:", volume_bench(torch, None, dev, 0, 1, 8)["ms_per_clip"])
torch.cuda.empty_cache()
print("again after empty_cache:", volume_bench(torch, None, dev, 0, 1, 8)["ms_per_clip"])
x = torch.rand(4, 2160, 3840, 3, device=dev); del x; torch.cuda.empty_cache()
print("after a 400 MB alloc/free:", volume_bench(torch, None, dev, 0, 1, 8)["ms_per_clip"])
frames_bench(torch, None, dev, 0, 1, False, 10); torch.cuda.empty_cache()
print("after frames_bench:", volume_bench(torch, None, dev, 0, 1, 8)["ms_per_clip"])
from dspfun_amd.engine import Stream
s = [Stream(), Stream()]
print("with two library streams alive:", volume_bench(torch, None, dev, 0, 1, 8)["ms_per_clip"])
