"""Which kernels a chirp-z zoom frame spends its time in (1080p -> 3.7x centered, and config 3's shape off the grid): run under
    rocprofv3 --kernel-trace --stats -d gpurun_out/cztprof -o z -- python3 tools/zoom_czt_probe.py [frames]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd.zoom import Zoom
w, h = 1920, 1080
z = Zoom(torch, torch.rand(h, w, 3, device="cuda:0"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for _ in range(n):
    z.frame(int(w * 3.7), int(h * 3.7), (3.7, 1.0), (3.7, 1.0), 12.5, -4.25, 1, method="czt")
torch.cuda.synchronize()
