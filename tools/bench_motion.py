"""BASELINE config 5 (motion on a 1920x1080x256 yuv420p clip, frame-batch sharded): every rank owns a contiguous range of
frames (dspfun_amd.dist.shard_range) of all three planes and runs motion's per-frame block loop on them -- 8-bit load,
2-D DCT-II, uniform scaling, quantiser, DCT-III, 8-bit store (motion/motion.c:617-776) -- through the fused
dspfft_execute_roundtrip_u8.  Frames are independent: no collective on the data path; the ranks only agree on the time.

    python tools/bench_motion.py                                   # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_motion.py
    ... tools/bench_motion.py volume     # `-b 0x0x0`: ONE 3-D block over the whole clip per plane (dspfun_amd.dist.SlabDCT3D: local y, x
                                         # passes, all-to-all over RCCL/xGMI in row pieces that overlap the passes, local z pass; float
                                         # forward + inverse, motion.c:535-552,641,753).  Unmeasured on more than one GPU until a SCALE record exists.
Prints one JSON line on rank 0."""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, REDFT10, REDFT01
from dspfun_amd.dist import shard_range

rank, local, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
dev = torch.device("cuda", local if world > 1 else 0)
torch.cuda.set_device(dev)
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=dev)

FRAMES, QUANT, REPS = 256, 20.0, 5      # motion --quant 20 (motion/README.md)
if len(sys.argv) > 1 and sys.argv[1] == "volume":
    from dspfun_amd.dist import SlabDCT3D
    chunks = int(os.environ.get("SLAB_CHUNKS", "4"))
    engs, vols = [], []
    for (h, w) in ((1080, 1920), (540, 960), (540, 960)):
        e = SlabDCT3D(FRAMES, h, w, chunks=chunks)
        engs.append(e)
        vols.append((torch.rand(e.dl, h, w, device=dev) * 255).floor())

    def clip3d():
        errs = []
        for e, v in zip(engs, vols):
            back = e.inverse(e.forward(v.clone()))
            errs.append(back)
        return errs

    def barrier3d():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    clip3d()
    barrier3d()
    t0 = time.perf_counter()
    for _ in range(REPS):
        outs = clip3d()
    barrier3d()
    dt = (time.perf_counter() - t0) / REPS
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    err = max(float((o - v).abs().max()) for o, v in zip(outs, vols) if v.numel())
    if rank == 0:
        samples = FRAMES * (1080 * 1920 + 2 * 540 * 960)
        print(json.dumps({"workload": "motion yuv420p 1920x1080x256, one 3-D block per plane (-b 0x0x0), float forward + inverse (incl. a clone of the input per plane)",
                          "n_gpus": world, "frames_per_rank": engs[0].dl, "row_pieces": engs[0].P, "ms_per_clip": round(dt * 1e3, 3),
                          "Msamples_per_s": round(samples / dt / 1e6), "algorithmic_GBps_per_gpu": round(samples * 16 / dt / 1e9 / world, 1),
                          "max_abs_roundtrip_error_0_255": err, "parallelism": f"slab x{world}: 2 all-to-alls per roundtrip (RCCL), pipelined in {engs[0].P} row pieces"}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(0)
lo, hi = shard_range(FRAMES, rank, world)
nf = hi - lo
r2 = math.sqrt(2.0)
planes = []
for (h, w) in ((1080, 1920), (540, 960), (540, 960)):            # Y, U, V of yuv420p (motion.c:61-67)
    # motion.c:644-647 with depth 1: 2 sqrt2 / sqrt2 (the z index is always 0), 1/sqrt2 more at x == 0 and at y == 0
    fwd = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=nf, idist=h * w, odist=h * w).set_scale(2.0)
    inv = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=nf, idist=h * w, odist=h * w, first_axis_first=True).set_scale(1.0 / 2.0 / (4.0 * h * w))
    for a in range(2):
        fwd.set_axis_scale0(a, 1.0, 1.0 / r2); inv.set_axis_scale0(a, r2, 1.0)
    src = (torch.rand(nf, h, w, device=dev) * 255).to(torch.uint8)
    planes.append(dict(h=h, w=w, fwd=fwd, inv=inv, src=src, dst=torch.empty_like(src), work=torch.empty(nf, h, w, device=dev),
                       flt=dict(active=(1, h, w), minbuf_hw=(h, w), block_depth=1, band_begin=(0, 0, 0), band_end=(1, h, w), quantizer=QUANT * 8 * math.sqrt(w * h))))     # motion.c:570
coded = torch.zeros(1, dtype=torch.int64, device=dev)

def clip():
    for p in planes:
        p["fwd"].roundtrip_u8(p["inv"], p["src"].data_ptr(), p["dst"].data_ptr(), p["work"].data_ptr(), 1.0, filter=p["flt"], d_coded=coded.data_ptr())

def barrier():
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()

clip()
barrier()
t0 = time.perf_counter()
for _ in range(REPS):
    clip()
barrier()
dt = (time.perf_counter() - t0) / REPS
if dist is not None:
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
err = max(int((p["dst"].int() - p["src"].int()).abs().max()) for p in planes)
if rank == 0:
    samples = FRAMES * (1080 * 1920 + 2 * 540 * 960)
    print(json.dumps({"workload": "motion yuv420p 1920x1080x256, per-frame blocks, --quant 20, u8 in -> u8 out", "n_gpus": world, "frames_per_rank": nf,
                      "ms_per_clip": round(dt * 1e3, 3), "frames_per_s": round(FRAMES / dt), "Msamples_per_s": round(samples / dt / 1e6),
                      "algorithmic_GBps_per_gpu": round(samples * 18 / dt / 1e9 / world, 1), "max_abs_u8_change": err,
                      "parallelism": f"frame-sharded x{world}, no collective"}))
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
