"""GPU (-m gpu): dspfft_execute_roundtrip_u8 over a clip in slices (engine.cpp roundtrip_sliced) against the same clip in three launches, through the C ABI:
bytes and the count of coded coefficients equal -- with the quantiser alone (motion --quant), with the position-dependent filter (band, damp / boost, threshold,
DC rule) and with a filter block depth > 1, where a slice must start on a block boundary because the filter finds a frame's place in its block from its offset in
the work area (motion/motion.c:591,613-615,683-744).  The switches are read once per process: every setting is a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = r'''
import math, sys, zlib
sys.path.insert(0, %(root)r)
import torch
from dspfun_amd import Plan, REDFT10, REDFT01
frames, h, w, bd, general = %(frames)d, 1080, %(w)d, %(bd)d, %(general)d
dev = torch.device("cuda", 0)
r2 = math.sqrt(2.0)
fwd = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=frames, idist=h * w, odist=h * w).set_scale(2.0)
inv = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=frames, idist=h * w, odist=h * w, first_axis_first=True).set_scale(1.0 / 2.0 / (4.0 * h * w))
for a in range(2):
    fwd.set_axis_scale0(a, 1.0, 1.0 / r2); inv.set_axis_scale0(a, r2, 1.0)
g = torch.Generator(device=dev); g.manual_seed(11)
src = torch.randint(0, 256, (frames, h, w), dtype=torch.uint8, device=dev, generator=g)
dst = torch.zeros_like(src)
work = torch.empty(frames, h, w, device=dev)
coded = torch.zeros(1, dtype=torch.int64, device=dev)
flt = dict(active=(bd, h, w), minbuf_hw=(h, w), block_depth=bd, band_begin=(0, 0, 0), band_end=(bd, h, w), quantizer=20.0 * 8 * math.sqrt(w * h))
if general:
    flt.update(band_begin=(1 if bd > 1 else 0, 2, 3), band_end=(bd, h - 100, w - 7), damp=0.5, boost=1.25, threshold_lo=0.0, threshold_hi=3.0e7, preserve_dc=1)
fwd.roundtrip_u8(inv, src.data_ptr(), dst.data_ptr(), work.data_ptr(), 1.0, filter=flt, d_coded=coded.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
d = fwd.describe()
print("RESULT", "%%08x" %% zlib.crc32(dst.cpu().numpy().tobytes()), int(coded.item()), "sliced" if "roundtrip_u8 in slices of" in d else "whole",
      d.split("roundtrip_u8 in slices of")[-1].split(":")[0].strip().replace(" ", "_") if "roundtrip_u8 in slices of" in d else "-")
'''


def run(env, **kw):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=os.path.dirname(HERE), **kw)], env=e, capture_output=True, text=True, timeout=900)
    lines = [x for x in r.stdout.splitlines() if x.startswith("RESULT")]
    assert lines, r.stderr[-2000:]
    return lines[0].split()[1:]


@pytest.mark.parametrize("w,frames,bd,general", [(1920, 64, 1, 0), (960, 40, 1, 1), (960, 48, 4, 1)])
def test_sliced_clip_is_the_three_launch_clip(w, frames, bd, general):
    kw = dict(w=w, frames=frames, bd=bd, general=general)
    whole = run({"DSPFFT_RT_SLICE": "0"}, **kw)
    assert whole[2] == "whole" and int(whole[1]) > 0
    for env in ({}, {"DSPFFT_RT_SLICE": "10", "DSPFFT_RT_STREAMS": "2"}, {"DSPFFT_RT_SLICE": "7", "DSPFFT_RT_STREAMS": "1"}):
        got = run(env, **kw)
        assert got[2] == "sliced", (env, got)
        assert got[:2] == whole[:2], (env, got, whole)
        if bd > 1:                       # slices start on block boundaries
            assert int(got[3].split("_")[0]) % bd == 0, got
