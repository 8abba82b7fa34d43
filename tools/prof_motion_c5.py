"""BASELINE config 5's per-frame clip (Y, U, V planes, 256 frames, u8 -> u8 with --quant 20) a few times over, for rocprofv3 --kernel-trace --stats /
--pmc: the same calls as tools/bench_motion.py frames_bench.   REPS=5 python3 tools/prof_motion_c5.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_motion as bm
dev = torch.device("cuda", 0)
print(bm.frames_bench(torch, None, dev, 0, 1, False, int(os.environ.get("REPS", "5"))))
