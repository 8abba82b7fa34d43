"""motion config 5, per-frame blocks u8 -> u8 (motion/motion.c:613-615,617-776): the clip walked in slices of S frames whose float
intermediate the 256 MB Infinity Cache holds, against the whole clip in three launches.  Every slice reuses ONE float work buffer of S
frames, so the intermediate never has to reach HBM.  Rows: plane, column tile width K, slice size S, ms per clip (HIP events, REPS clips).

    python tools/motion_chunks.py [luma|chroma] [reps]
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

FRAMES, QUANT = 256, 20.0


def plans(h, w, nf):
    from dspfun_amd import Plan, REDFT10, REDFT01
    r2 = math.sqrt(2.0)
    fwd = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=nf, idist=h * w, odist=h * w).set_scale(2.0)
    inv = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=nf, idist=h * w, odist=h * w, first_axis_first=True).set_scale(1.0 / 2.0 / (4.0 * h * w))
    for a in range(2):
        fwd.set_axis_scale0(a, 1.0, 1.0 / r2); inv.set_axis_scale0(a, r2, 1.0)
    return fwd, inv


def run(h, w, kpref, S, reps, src, dst, ref=None, ns=1):
    dev = src.device
    if kpref:
        os.environ["DSPFFT_COL_KPREF"] = str(kpref)
    else:
        os.environ.pop("DSPFFT_COL_KPREF", None)
    fwd, inv = plans(h, w, S)
    rem = FRAMES % S
    fr, ir = plans(h, w, rem) if rem else (None, None)
    works = [torch.empty(S, h, w, device=dev) for _ in range(ns)]
    side = [torch.cuda.Stream() for _ in range(ns)] if ns > 1 else []
    flt = dict(active=(1, h, w), minbuf_hw=(h, w), block_depth=1, band_begin=(0, 0, 0), band_end=(1, h, w), quantizer=QUANT * 8 * math.sqrt(w * h))
    coded = torch.zeros(1, dtype=torch.int64, device=dev)
    cur = torch.cuda.current_stream()
    fs = h * w

    def clip():
        if ns > 1:
            ev = torch.cuda.Event(); ev.record(cur)
            for s_ in side:
                s_.wait_event(ev)
        for c in range(FRAMES // S):
            st = side[c % ns].cuda_stream if ns > 1 else cur.cuda_stream
            fwd.roundtrip_u8(inv, src.data_ptr() + c * S * fs, dst.data_ptr() + c * S * fs, works[c % ns].data_ptr(), 1.0, filter=flt, d_coded=coded.data_ptr(), stream=st)
        if rem:
            c = FRAMES // S
            st = side[c % ns].cuda_stream if ns > 1 else cur.cuda_stream
            fr.roundtrip_u8(ir, src.data_ptr() + c * S * fs, dst.data_ptr() + c * S * fs, works[c % ns].data_ptr(), 1.0, filter=flt, d_coded=coded.data_ptr(), stream=st)
        if ns > 1:
            for s_ in side:
                e2 = torch.cuda.Event(); e2.record(s_)
                cur.wait_event(e2)

    dst.zero_()
    clip(); clip()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        clip()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    same = None if ref is None else bool(torch.equal(dst, ref))
    desc = fwd.describe().splitlines()[-1][:60]
    print(f"{h}x{w} K={kpref or 'default'} streams={ns} S={S:4d} work={S * fs * 4 / 1e6:7.1f} MB  {ms:7.3f} ms/clip  identical_to_whole_clip={same}  [{desc}]", flush=True)
    return ms


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "luma"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    h, w = (1080, 1920) if what == "luma" else (540, 960)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(5)
    src = torch.randint(0, 256, (FRAMES, h, w), dtype=torch.uint8, device=dev, generator=g)
    dst = torch.empty_like(src)
    run(h, w, 0, FRAMES, reps, src, dst)
    ref = dst.clone()
    if len(sys.argv) > 3:            # one configuration (profiling runs): K S streams
        run(h, w, int(sys.argv[3]), int(sys.argv[4]), reps, src, dst, ref, int(sys.argv[5]))
        return
    ks = (0, 8) if what == "luma" else (0,)
    for ns in (1, 2, 3):
        for k in ks:
            for S in ((FRAMES, 128, 64, 48, 32, 24, 16, 12, 8, 4) if ns == 1 else (32, 24, 16, 12, 8, 6, 4) if what == "luma" else (128, 64, 32, 16)):
                run(h, w, k, S, reps, src, dst, ref, ns)


if __name__ == "__main__":
    main()
