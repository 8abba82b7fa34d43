"""Cross-check of spec_list.h's entries with the planner's timed search: with the listed kernels switched off (DSPFFT_NO_SPEC=1) and
planning effort 2 (DSPFFT_JIT=1 DSPFFT_JIT_TUNE=1) the planner compiles and times its own candidates for the same frame sizes."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01

def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

for (h, w, c) in [(1080, 1920, 3), (720, 1280, 3), (1440, 2560, 3), (2160, 3840, 3), (1080, 1920, 1), (540, 960, 3), (2160, 4096, 3)]:
    x = torch.rand(h, w, c, device="cuda:0")
    row = {"size": f"{w}x{h}x{c}"}
    for tag, env in (("listed", {}), ("searched", {"DSPFFT_NO_SPEC": "1", "DSPFFT_JIT": "1", "DSPFFT_JIT_TUNE": "1"})):
        for k in ("DSPFFT_NO_SPEC", "DSPFFT_JIT", "DSPFFT_JIT_TUNE"): os.environ.pop(k, None)
        os.environ.update(env)
        f = Plan.image(h, w, c, REDFT10); i = Plan.image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * h * w))
        row[tag + "_us"] = round(t(lambda: (f.execute(x.data_ptr()), i.execute(x.data_ptr()))) * 1000, 1)
        row[tag] = [l.split(": ", 1)[1][:75] for l in f.describe().splitlines()[1:3]]
    print(json.dumps(row), flush=True)
