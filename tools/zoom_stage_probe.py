"""Which kernels a config-3 zoom frame by fast transforms spends its time in: run under
    rocprofv3 --kernel-trace --stats -d gpurun_out/zoomprof -o z -- python3 tools/zoom_stage_probe.py
(DSPFFT_ZOOM_ORDER=y: the column pass last, round 3's first cut; DSPFFT_ZOOM_XROWS=0: round 3's two-transform row pass).  An argument = the number
of frames (12; 600 for averages at settled clocks)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd.zoom import Zoom
w, h = 1920, 1080
z = Zoom(torch, torch.rand(h, w, 3, device="cuda:0"))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), method="fft")
torch.cuda.synchronize()
