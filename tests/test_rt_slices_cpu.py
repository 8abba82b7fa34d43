"""CPU: dspfft_execute_roundtrip_u8 over a clip in slices (engine.cpp roundtrip_sliced: slice plans on the narrow column tile, one reused work area
per stream, a remainder slice) gives the bytes and the count of coded coefficients of the whole clip in three launches -- motion's per-frame
blocks are independent (motion/motion.c:591,613-615).  Through the test-only emulation library; the switches are read once per process, so each
setting runs in a child."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = r'''
import math, sys, zlib
import numpy as np
sys.path.insert(0, %(here)r); sys.path.insert(0, %(root)r)
import ctypes as C
from emul_lib import emul
from dspfun_amd import Plan, REDFT10, REDFT01
import oracle_lib as ol
L = emul()
frames, h, w = 5, 1080, 960
r2 = math.sqrt(2.0)
fwd = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=frames, idist=h * w, odist=h * w, lib=L).set_scale(2.0)
inv = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=frames, idist=h * w, odist=h * w, first_axis_first=True, lib=L).set_scale(1.0 / 2.0 / (4.0 * h * w))
for a in range(2):
    fwd.set_axis_scale0(a, 1.0, 1.0 / r2); inv.set_axis_scale0(a, r2, 1.0)
src = ol.synth_u8(0xD5F0005, frames * h * w)
dst = np.zeros_like(src)
work = np.zeros(frames * h * w, dtype=np.float32)
coded = np.zeros(1, dtype=np.uint64)
flt = dict(active=(1, h, w), minbuf_hw=(h, w), block_depth=1, band_begin=(0, 0, 0), band_end=(1, h, w), quantizer=20.0 * 8 * math.sqrt(w * h))
fwd.roundtrip_u8(inv, src.ctypes.data, dst.ctypes.data, work.ctypes.data, 1.0, filter=flt, d_coded=coded.ctypes.data)
print("RESULT", "%%08x" %% zlib.crc32(dst.tobytes()), int(coded[0]), int(np.abs(dst.astype(int) - src.astype(int)).max()), "sliced" if "roundtrip_u8 in slices of" in fwd.describe() and "K=8" in fwd.describe().split("roundtrip_u8 in slices of")[-1] else "whole", fwd.describe().split("roundtrip_u8 in slices of")[-1][:40].replace(" ", "_"))
'''


def run(env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % {"here": HERE, "root": os.path.dirname(HERE)}], env=e, capture_output=True, text=True, timeout=900)
    lines = [x for x in r.stdout.splitlines() if x.startswith("RESULT")]
    assert lines, r.stderr[-2000:]
    return lines[0].split()[1:]


def test_sliced_clip_is_the_whole_clip():
    whole = run({"DSPFFT_RT_SLICE": "0"})
    assert int(whole[2]) < 64 and int(whole[1]) > 0              # (a quantised roundtrip: close to the input, some coefficients coded)
    for env in ({"DSPFFT_RT_SLICE": "2", "DSPFFT_RT_STREAMS": "2"}, {"DSPFFT_RT_SLICE": "2", "DSPFFT_RT_STREAMS": "1"}, {"DSPFFT_RT_SLICE": "3", "DSPFFT_RT_STREAMS": "2"}):
        got = run(env)
        assert got[:3] == whole[:3], (env, got, whole)
        assert got[3] == "sliced" and whole[3] == "whole", (got, whole)
        assert got[4].startswith("_%s_frames_(last:_%d)_on_%s_stream" % (env["DSPFFT_RT_SLICE"], 5 % int(env["DSPFFT_RT_SLICE"]) or int(env["DSPFFT_RT_SLICE"]), env["DSPFFT_RT_STREAMS"])), got
