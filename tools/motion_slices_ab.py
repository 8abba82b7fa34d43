"""motion config 5, per-frame blocks u8 -> u8 through dspfft_execute_roundtrip_u8 as bench.py's motion_c5 calls it, under the library's own slicing
(engine.cpp roundtrip_sliced).  The switches are read once per process, so every row is a child process:

    python tools/motion_slices_ab.py            # rows: DSPFFT_RT_SLICE / DSPFFT_RT_STREAMS settings x {luma clip, whole Y+U+V clip}, ms and a checksum
"""
import math
import os
import subprocess
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FRAMES, QUANT = int(os.environ.get("FRAMES", "256")), 20.0
PLANES = ((1080, 1920), (540, 960), (540, 960))


def child(reps):
    import torch
    from dspfun_amd import Plan, REDFT10, REDFT01
    dev = torch.device("cuda", 0)
    r2 = math.sqrt(2.0)
    planes = []
    g = torch.Generator(device=dev); g.manual_seed(5)
    for (h, w) in PLANES:
        fwd = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=FRAMES, idist=h * w, odist=h * w).set_scale(2.0)
        inv = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=FRAMES, idist=h * w, odist=h * w, first_axis_first=True).set_scale(1.0 / 2.0 / (4.0 * h * w))
        for a in range(2):
            fwd.set_axis_scale0(a, 1.0, 1.0 / r2); inv.set_axis_scale0(a, r2, 1.0)
        src = torch.randint(0, 256, (FRAMES, h, w), dtype=torch.uint8, device=dev, generator=g)
        planes.append(dict(fwd=fwd, inv=inv, src=src, dst=torch.zeros_like(src), work=torch.empty(FRAMES, h, w, device=dev),
                           flt=dict(active=(1, h, w), minbuf_hw=(h, w), block_depth=1, band_begin=(0, 0, 0), band_end=(1, h, w), quantizer=QUANT * 8 * math.sqrt(w * h))))
    coded = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run(ps):
        for p in ps:
            p["fwd"].roundtrip_u8(p["inv"], p["src"].data_ptr(), p["dst"].data_ptr(), p["work"].data_ptr(), 1.0, filter=p["flt"], d_coded=coded.data_ptr(), stream=stream)

    out = []
    for name, ps in (("luma", planes[:1]), ("clip", planes)):
        run(ps); run(ps)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            run(ps)
        b.record()
        torch.cuda.synchronize()
        out.append(f"{name} {a.elapsed_time(b) / reps:.3f} ms")
    coded.zero_(); run(planes); torch.cuda.synchronize()
    crc = 0
    for p in planes:
        crc = zlib.crc32(p["dst"].cpu().numpy().tobytes(), crc)
    print("  ".join(out), f" crc32 {crc:08x}  coded {int(coded.item())}", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child(int(sys.argv[2]))
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rows = [("whole clip (DSPFFT_RT_SLICE=0)", {"DSPFFT_RT_SLICE": "0"}), ("library default", {})]
    for streams in ("1", "2"):
        for S in ("6", "8", "12", "16", "24", "32"):
            rows.append((f"slice {S} frames, {streams} stream(s)", {"DSPFFT_RT_SLICE": S, "DSPFFT_RT_STREAMS": streams}))
    rows += [("whole clip again", {"DSPFFT_RT_SLICE": "0"}), ("library default again", {})]
    for name, env in rows:
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(reps)], env=e, capture_output=True, text=True)
        line = [x for x in r.stdout.splitlines() if "crc32" in x]
        print(f"{name:36s} {line[0] if line else 'FAILED: ' + r.stderr[-400:]}", flush=True)


if __name__ == "__main__":
    main()
