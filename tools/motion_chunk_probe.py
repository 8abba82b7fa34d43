"""motion's per-frame pipeline (u8 -> 2-D DCT-II -> quantiser -> DCT-III -> u8, 256 luma frames of 1920x1080) in chunks of frames whose
float work set stays in the 256 MB Infinity Cache: the same three launches per chunk, every chunk through the SAME work buffer.
    python tools/motion_chunk_probe.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspfun_amd import Plan, REDFT10, REDFT01

FRAMES, QUANT = 256, 20.0
dev = torch.device("cuda:0")
r2 = math.sqrt(2.0)


def pipeline(h, w, nf):
    fwd = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=nf, idist=h * w, odist=h * w).set_scale(2.0)
    inv = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=nf, idist=h * w, odist=h * w, first_axis_first=True).set_scale(1.0 / 2.0 / (4.0 * h * w))
    for a in range(2):
        fwd.set_axis_scale0(a, 1.0, 1.0 / r2); inv.set_axis_scale0(a, r2, 1.0)
    flt = dict(active=(1, h, w), minbuf_hw=(h, w), block_depth=1, band_begin=(0, 0, 0), band_end=(1, h, w), quantizer=QUANT * 8 * math.sqrt(w * h))
    return fwd, inv, flt


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for (h, w) in ((1080, 1920), (540, 960)):
    src = (torch.rand(FRAMES, h, w, device=dev) * 255).to(torch.uint8)
    ref = None
    for chunk in (256, 64, 32, 16, 8, 4):
        fwd, inv, flt = pipeline(h, w, chunk)
        dst = torch.empty_like(src)
        work = torch.empty(chunk, h, w, device=dev)
        coded = torch.zeros(1, dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def clip():
            for f0 in range(0, FRAMES, chunk):
                fwd.roundtrip_u8(inv, src[f0].data_ptr(), dst[f0].data_ptr(), work.data_ptr(), 1.0, filter=flt, d_coded=coded.data_ptr(), stream=st)
        ms = timed(clip)
        if ref is None:
            ref = dst.clone()
        same = bool((dst == ref).all())
        print(f"{w}x{h} x {FRAMES} frames in chunks of {chunk:3d} (work set {chunk * h * w * 4 / 1e6:6.1f} MB): {ms:.3f} ms per clip, identical output: {same}")
