"""BASELINE config 5 (motion on a 1920x1080x256 yuv420p clip on 1/2/4/8 GPUs), as functions bench.py calls for its `motion_c5`
object and as a script:

    python tools/bench_motion.py                                   # one GPU: every mode below
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_motion.py [frames|volume]

* per-frame blocks (motion's default `-b 0x0x1`, motion/motion.c:174,613-615): every rank owns a contiguous range of frames
  (dspfun_amd.dist.shard_range) of all three planes and runs motion's per-frame block loop on them -- 8-bit load, 2-D DCT-II,
  uniform scaling, quantiser, DCT-III, 8-bit store (motion/motion.c:617-776) -- through the fused dspfft_execute_roundtrip_u8.
  Frames are independent: NO collective on the data path; the ranks only agree on the time.  Reported STRONG (one 256-frame clip
  split over the ranks) and WEAK (a 256-frame clip per rank), each over enough clips per timed region that eight ranks are not
  launch-bound (round 2 timed 5 clips: 0.66 ms and nine launches per rank and clip at 8 ranks).
* one 3-D block over the whole clip per plane (`-b 0x0x0`, motion.c:535-552,641,753): dspfun_amd.dist.SlabDCT3D -- local y and x
  passes, all-to-all over RCCL/xGMI in row pieces that overlap the passes, local z pass; float forward + inverse.  The only path
  with a collective.  Unmeasured on more than one GPU until a SCALE record exists.
Every timed region is bracketed by barrier + synchronize on both sides and the MAX over ranks is reported."""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FRAMES, QUANT = 256, 20.0                                       # motion --quant 20 (motion/README.md)
PLANES = ((1080, 1920), (540, 960), (540, 960))                 # Y, U, V of yuv420p (motion.c:61-67)
SAMPLES = FRAMES * sum(h * w for h, w in PLANES)
XGMI_PEAK_PER_GPU = 7 * 76.8e9                                  # MI355X: seven xGMI links per GPU, 76.8 GB/s per direction each


def _on_gpu(torch, dev):
    return torch.device(dev).type == "cuda"


def _barrier(torch, dist, dev="cuda"):
    if _on_gpu(torch, dev):
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if _on_gpu(torch, dev):
        torch.cuda.synchronize()


def _timed(torch, dist, dev, fn, reps):
    fn()
    _barrier(torch, dist, dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    _barrier(torch, dist, dev)
    dt = (time.perf_counter() - t0) / reps
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def frames_bench(torch, dist, dev, rank, world, weak, reps, frames=FRAMES, planes_hw=PLANES, lib=None):
    """per-frame blocks; weak: `frames` frames per rank, strong: `frames` frames over all ranks.
    frames, planes_hw, lib, dev = cpu: the CPU test of this very function (tests/test_dist_cpu.py: eight gloo ranks, a small clip, the emulation
    library) -- bench.py and the script below use the defaults"""
    from dspfun_amd import Plan, REDFT10, REDFT01
    from dspfun_amd.dist import shard_range
    FRAMES, PLANES = frames, planes_hw
    SAMPLES = FRAMES * sum(h * w for h, w in PLANES)
    lo, hi = (0, FRAMES) if weak else shard_range(FRAMES, rank, world)
    nf = hi - lo
    r2 = math.sqrt(2.0)
    planes = []
    for (h, w) in PLANES:
        if not nf:
            continue
        # motion.c:644-647 with depth 1: 2 sqrt2 / sqrt2 (the z index is always 0), 1/sqrt2 more at x == 0 and at y == 0
        fwd = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=nf, idist=h * w, odist=h * w, lib=lib).set_scale(2.0)
        inv = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=nf, idist=h * w, odist=h * w, first_axis_first=True, lib=lib).set_scale(1.0 / 2.0 / (4.0 * h * w))
        for a in range(2):
            fwd.set_axis_scale0(a, 1.0, 1.0 / r2); inv.set_axis_scale0(a, r2, 1.0)
        src = (torch.rand(nf, h, w, device=dev) * 255).to(torch.uint8)
        planes.append(dict(fwd=fwd, inv=inv, src=src, dst=torch.empty_like(src), work=torch.empty(nf, h, w, device=dev),
                           flt=dict(active=(1, h, w), minbuf_hw=(h, w), block_depth=1, band_begin=(0, 0, 0), band_end=(1, h, w),
                                    quantizer=QUANT * 8 * math.sqrt(w * h))))     # motion.c:570
    coded = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream if _on_gpu(torch, dev) else 0

    def clip():
        for p in planes:
            p["fwd"].roundtrip_u8(p["inv"], p["src"].data_ptr(), p["dst"].data_ptr(), p["work"].data_ptr(), 1.0, filter=p["flt"],
                                  d_coded=coded.data_ptr(), stream=stream)

    dt = _timed(torch, dist, dev, clip, reps)
    err = max([int((p["dst"].int() - p["src"].int()).abs().max()) for p in planes] or [0])
    clips = world if weak else 1
    return {"scaling": "weak" if weak else "strong", "frames_per_rank": nf, "clips_timed": reps, "ms_per_clip_round": round(dt * 1e3, 3),
            "frames_per_s": round(clips * FRAMES / dt), "Msamples_per_s": round(clips * SAMPLES / dt / 1e6),
            "algorithmic_GBps_per_gpu": round(clips * SAMPLES * 18 / dt / 1e9 / world, 1),        # 18 B/sample: u8 in + 16 (float roundtrip) + u8 out
            "max_abs_u8_change": err, "parallelism": f"frame-sharded x{world}, no collective"}


def volume_bench(torch, dist, dev, rank, world, reps, chunks=None, frames=FRAMES, planes_hw=PLANES, lib=None):
    """one 3-D block per plane over the whole clip: SlabDCT3D forward + inverse (two RCCL all-to-alls per plane)"""
    from dspfun_amd.dist import SlabDCT3D
    FRAMES, PLANES = frames, planes_hw
    SAMPLES = FRAMES * sum(h * w for h, w in PLANES)
    chunks = int(os.environ.get("SLAB_CHUNKS", "4")) if chunks is None else chunks
    engs, vols = [], []
    for (h, w) in PLANES:
        e = SlabDCT3D(FRAMES, h, w, chunks=chunks, lib=lib)
        engs.append(e)
        vols.append((torch.rand(e.dl, h, w, device=dev) * 255).floor())
    # forward() overwrites its input, so every roundtrip transforms the frames the one before returned (round 6; before that each clip paid for a
    # clone of its 3.2 GB of frames inside the timed region: 1.4 of 9.5 ms).  The error below is the drift after ALL of them.
    outs = [v.clone() for v in vols]
    nround = [0]

    def clip3d():
        outs[:] = [e.inverse(e.forward(o)) for e, o in zip(engs, outs)]
        nround[0] += 1

    # two timed regions, the faster one reported: the first region of this path was seen 50 % slow on some runs (9.9 vs 15.2 ms per clip,
    # same binary; the caching allocator still carving its blocks), the second never
    dt = min(_timed(torch, dist, dev, clip3d, max(1, reps // 2)), _timed(torch, dist, dev, clip3d, reps))
    err = max([float((o - v).abs().max()) for o, v in zip(outs, vols) if v.numel()] or [0.0])
    if dist is not None:
        t = torch.tensor([err], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        err = float(t.item())
    # bytes one rank sends per all-to-all: its share of the volume minus what stays local
    a2a = SAMPLES * 4 / world * (world - 1) / world if world > 1 else 0
    # the exchange against the xGMI roofline.  The exchanges of ONE direction of the three planes (half of a roundtrip's), alone on
    # the links and each waited for: time, bytes that left this rank, and the fraction of 7 links x 76.8 GB/s per direction they ran at.  With
    # one rank there is no exchange: null.
    exchange_ms = xgmi_gbps = xgmi_frac = exchange_bytes = None
    if world > 1:
        sent = [0]

        def exchanges():
            sent[0] = sum(e.exchange_alone(v) for e, v in zip(engs, vols))
        dtx = _timed(torch, dist, dev, exchanges, max(2, reps))
        exchange_ms = round(dtx * 1e3, 4)
        exchange_bytes = int(sent[0])
        xgmi_gbps = float(f"{sent[0] / dtx / 1e9:.6g}")
        xgmi_frac = float(f"{sent[0] / dtx / XGMI_PEAK_PER_GPU:.6g}")
    return {"scaling": "strong", "frames_per_rank": engs[0].dl, "row_pieces": engs[0].P, "clips_timed": reps, "ms_per_clip": round(dt * 1e3, 3),
            "Msamples_per_s": round(SAMPLES / dt / 1e6), "algorithmic_GBps_per_gpu": round(SAMPLES * 16 / dt / 1e9 / world, 1),
            "alltoall_MB_sent_per_rank_per_exchange": round(a2a / 1e6, 1), "exchanges_per_clip": 2 * len(PLANES) if world > 1 else 0,
            "exchange_ms": exchange_ms, "exchange_bytes_sent_per_rank": exchange_bytes, "xgmi_GBps_sent_per_rank": xgmi_gbps, "xgmi_frac": xgmi_frac,
            "exchange_note": "exchange_ms = the all-to-alls of ONE direction of the three planes (half of a clip's), alone on the links; xgmi_frac = bytes this rank "
                             "sent to the other ranks / exchange time / (7 links x 76.8 GB/s per direction); null with one rank (no exchange)",
            "max_abs_roundtrip_error_0_255": err, "roundtrips_behind_that_error": nround[0],
            "parallelism": f"slab x{world}: 2 all-to-alls per plane roundtrip (RCCL), pipelined in {engs[0].P} row pieces"}


def motion_c5(torch, dist, dev, rank, world, reps_frames=50, reps_volume=8, frames=FRAMES, planes_hw=PLANES, lib=None):
    """the object bench.py attaches to its JSON line (every rank must call it: the volume mode is collective).
    frames, planes_hw, lib (and dev = cpu): see frames_bench"""
    kw = dict(frames=frames, planes_hw=planes_hw, lib=lib)
    h0, w0 = planes_hw[0]

    def trim():
        if _on_gpu(torch, dev):
            torch.cuda.empty_cache()
    out = {"workload": f"motion on a {w0}x{h0}x{frames} yuv420p clip (BASELINE configs[4]: 1920x1080x256); per-frame blocks u8 -> u8 with --quant 20, and one 3-D "
                       "block per plane (-b 0x0x0) float forward + inverse through SlabDCT3D",
           "per_frame_strong": frames_bench(torch, dist, dev, rank, world, False, reps_frames, **kw)}
    trim()
    out["per_frame_weak"] = frames_bench(torch, dist, dev, rank, world, True, max(4, reps_frames // 4), **kw) if world > 1 else "n_gpus = 1: same as per_frame_strong"
    trim()
    try:
        out["volume_3d"] = volume_bench(torch, dist, dev, rank, world, reps_volume, **kw)
    except Exception as e:              # the collective path must not take the frame-sharded numbers down with it
        out["volume_3d"] = {"error": f"{type(e).__name__}: {e}"}
    trim()
    return out


def main():
    import torch
    rank, local, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "volume":
        res = volume_bench(torch, dist, dev, rank, world, 8)
    elif what == "frames":
        res = {"strong": frames_bench(torch, dist, dev, rank, world, False, 50), "weak": frames_bench(torch, dist, dev, rank, world, True, 12)}
    else:
        res = motion_c5(torch, dist, dev, rank, world)
    if rank == 0:
        res["n_gpus"] = world
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
