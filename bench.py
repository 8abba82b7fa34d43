#!/usr/bin/env python3
"""bench.py -- headline benchmark (BASELINE.json): Mpixels/s of the DCT-II + DCT-III roundtrip on
3840x2160x3 f32 frames, device-resident, as a fraction of the MI355X HBM roofline.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ...`, one rank per GPU)

A "step" = one roundtrip (2-D REDFT10 in place, then 2-D REDFT01 in place with the 1/(4wh) gain
fused) over a batch of FRAMES independent frames per GPU -- the per-frame plan of spec/spec.c:63 +
spec/ispec.c:165, and what motion's default `-b 0x0x1` runs per frame (motion/motion.c:174,613-753).
Frames are independent units, so ranks shard them with no data-path collective (weak scaling).

One JSON line on stdout (rank 0).  roofline.* describes the dominant kernel, timed with HIP events on
the launch stream (torch's current stream -- the library launches on the stream handle it is given).
cpu_baseline is the oracle's O(N log N) CPU port (oracle/cpu_port.c, "kind": "port"), timed on the
host cores of the same box on a bounded sample; it is a reported baseline, not the target.
"""
import argparse
import json
import threading
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, C = 2160, 3840, 3
ALG_BYTES_PER_PIXEL = 48.0        # 16 B/sample roundtrip = 48 B/pixel
HBM_PEAK = 8.0e12                 # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)
SEED = 0xD5F0002
DRIFT_PER_ROUNDTRIP = 2e-6        # growth of |frame - input| per in-place roundtrip allowed in a run (measured: 8.4e-7; ONE roundtrip is pinned at
DRIFT_BOUND = 1e-3                # <= 5e-6 by tests/test_gpu_parity.py): the sanity bound is max(DRIFT_BOUND, n * DRIFT_PER_ROUNDTRIP) -- set in main()


def synth_frames(torch, nframes, device):
    """splitmix64-seeded uniform [0,1) f32 (state advances once per sample,
    f32 = (u >> 40) * 2^-24), generated on the host and copied to the device."""
    import numpy as np
    n = H * W * C
    out = torch.empty((nframes, H, W, C), dtype=torch.float32, device=device)
    idx = np.arange(1, n + 1, dtype=np.uint64)
    for f in range(nframes):
        with np.errstate(over="ignore"):
            z = np.uint64(SEED + f) + idx * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        x = (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
        out[f] = torch.from_numpy(x.reshape(H, W, C)).to(device)
    return out


def physical_core_cpus():
    """one logical CPU per physical core that this process may run on, grouped by socket (package) in ascending order: [cpu, ...], {package: count}.
    From /sys/devices/system/cpu/cpuN/topology; falls back to the affinity mask as it is."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    seen, cpus, per_pkg = set(), [], {}
    for cpu in allowed:
        base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
        try:
            with open(base + "core_id") as f:
                core = int(f.read())
            with open(base + "physical_package_id") as f:
                pkg = int(f.read())
        except (OSError, ValueError):
            core, pkg = cpu, 0
        if (pkg, core) in seen:
            continue
        seen.add((pkg, core))
        cpus.append((pkg, cpu))
        per_pkg[pkg] = per_pkg.get(pkg, 0) + 1
    cpus.sort()
    return [c for _, c in cpus], per_pkg


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited.  The GPU boxes' containers see 256 logical
    CPUs and are held to 16 by the CFS quota: threads beyond it are throttled, not run (profiles/r06_cpu_scale.txt: 16 threads 178-188 Mpix/s,
    128 threads 54-62)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(max_seconds=30.0):
    """Oracle CPU port (f32) on full 3840x2160x3 roundtrips: ONE thread and ALL physical cores (SURVEY.md 8d asks for both).  Only this leg of
    bench.py touches oracle/.  The all-core figure runs oracle/cpu_port_impl.h's bench entry (round 5's all-core number was
    2.2x the single-thread one -- a frame first touched by the caller's thread, plans rebuilt and thread teams restarted in every pass): one
    thread team for the whole run, each thread pinned to its own physical core, the frame first touched by the threads that transform its
    rows, static row / column-block ranges, plans built before the clock starts (as FFTW's are).  Never called FFTW."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    import ctypes as C_
    L = ol.lib()
    x = ol.synth_f32(SEED, H * W * C)
    cpus, per_pkg = physical_core_cpus()
    quota = cpu_quota()
    usable = len(cpus) if quota is None else max(1, min(len(cpus), int(quota + 0.5)))      # "all cores" = all the container is given time on
    cpus = cpus[:max(1, min(usable, L.cpu_port_max_threads()))]
    fn = L.cpu_port_roundtrip_bench_f32
    fn.restype = C_.c_int
    fn.argtypes = [C_.c_int, C_.c_int, C_.c_int, C_.c_void_p, C_.c_int, C_.c_void_p, C_.c_int, C_.POINTER(C_.c_double), C_.POINTER(C_.c_double)]

    def run(nthr, budget):
        """a probe of one roundtrip sizes the timed run to the budget"""
        pin = (C_.c_int * nthr)(*cpus[:nthr]) if nthr > 1 else None
        sec, err = C_.c_double(0), C_.c_double(0)
        rc = fn(H, W, C, x.ctypes.data, nthr, pin, 1, C_.byref(sec), C_.byref(err))
        pinned = rc == 0
        if rc == -3:                         # affinity refused (a container's cpuset): unpinned
            pin = None
            rc = fn(H, W, C, x.ctypes.data, nthr, pin, 1, C_.byref(sec), C_.byref(err))
        assert rc == 0, rc
        reps = int(max(1, min(256, (budget - 2 * sec.value) / max(sec.value, 1e-4))))
        rc = fn(H, W, C, x.ctypes.data, nthr, pin, reps, C_.byref(sec), C_.byref(err))
        assert rc == 0, rc
        return reps, sec.value, err.value, pinned and pin is not None

    r1, e1, err1, _ = run(1, max_seconds * 0.4)
    ra, ea, erra, pinned = run(len(cpus), max_seconds * 0.4)
    return {"value": round(ra * H * W / 1e6 / ea, 3), "unit": "Mpixels/s", "cores": len(cpus), "kind": "port",
            "single_thread_value": round(r1 * H * W / 1e6 / e1, 3), "all_core_over_single_thread": round((ra / ea) / (r1 / e1), 1),
            "cpu_quota": quota, "physical_cores": sum(per_pkg.values()),
            "sample": f"{ra} consecutive in-place roundtrip(s) of one 3840x2160x3 f32 frame on {len(cpus)} OpenMP threads, "
                      f"{'each pinned to its own physical core' if pinned else 'unpinned (affinity not settable here)'} "
                      f"(the box: {', '.join(f'socket {k}: {v} cores' for k, v in sorted(per_pkg.items()))}, {os.cpu_count()} logical CPUs; "
                      f"{'no CPU quota' if quota is None else f'the container is held to {quota:g} CPUs of time by its cgroup quota, so that many threads: more are throttled, not run'}), "
                      f"frame first touched by the threads that own its rows, static row / 16-column-block ranges, one thread team, plans built before the clock starts; "
                      f"{r1} roundtrip(s) with one thread; oracle/cpu_port_impl.h cpu_port_roundtrip_bench (max abs err after the last roundtrip {max(err1, erra):.1e})"}


def scipy_cpu_baseline(max_seconds=10.0):
    """scipy's pocketfft on the same frame, all host threads -- an INDEPENDENT CPU number beside the port's (never called
    FFTW, never the oracle).  None when scipy does not import on this box."""
    try:
        import numpy as np
        import scipy
        import scipy.fft as sf
    except Exception:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol                      # only for the synthetic frame of SURVEY 8d
    x = ol.synth_f32(SEED, H * W * C).reshape(H, W, C)
    reps, t0 = 0, time.perf_counter()
    while True:
        f = sf.dctn(x, type=2, axes=(0, 1), workers=-1)
        b = sf.dctn(f, type=3, axes=(0, 1), workers=-1)
        reps += 1
        el = time.perf_counter() - t0
        if el * (reps + 1) / reps > max_seconds or reps >= 32:
            break
    err = float(np.abs(b / np.float32(4.0 * W * H) - x).max())
    return {"value": round(reps * H * W / 1e6 / el, 3), "unit": "Mpixels/s", "cores": os.cpu_count(), "kind": "scipy.fft.dctn (pocketfft), not FFTW",
            "sample": f"{reps} roundtrip(s) of one 3840x2160x3 f32 frame, scipy {scipy.__version__}, workers=-1 (roundtrip max abs err {err:.1e})"}


def fftw_cpu_baseline(max_seconds=20.0):
    """The real FFTW (north_star "next to the FFTW CPU path") when the box has it: dlopen libfftw3f.so.3
    (never our own alias, which has no .3 soname and is not on the loader path), fftwf_plan_many_r2r(FFTW_ESTIMATE) of the 4K
    roundtrip with 1 thread and with all cores.  Returns None when FFTW is absent (this image and, so far, the GPU boxes)."""
    import ctypes as C
    try:
        fw = C.CDLL("libfftw3f.so.3")
        if not hasattr(fw, "fftwf_plan_many_r2r") or not hasattr(fw, "fftwf_version"):
            return None
    except OSError:
        return None
    import numpy as np
    ip = C.POINTER(C.c_int)
    fw.fftwf_plan_many_r2r.restype = C.c_void_p
    fw.fftwf_plan_many_r2r.argtypes = [C.c_int, ip, C.c_int, C.c_void_p, ip, C.c_int, C.c_int, C.c_void_p, ip, C.c_int, C.c_int, ip, C.c_uint]
    fw.fftwf_execute.argtypes = [C.c_void_p]
    fw.fftwf_destroy_plan.argtypes = [C.c_void_p]
    threads_ok = False
    try:
        ft = C.CDLL("libfftw3f_threads.so.3")
        threads_ok = bool(ft.fftwf_init_threads())
    except OSError:
        ft = None
    x = (np.random.default_rng(SEED).random(H * W * C, dtype=np.float32))
    ia = lambda v: (C.c_int * len(v))(*v)
    out = {}
    for label, nthr in (("1", 1), ("all", os.cpu_count() or 1)):
        if nthr > 1 and not threads_ok:
            continue
        if threads_ok:
            ft.fftwf_plan_with_nthreads(nthr)
        pf = fw.fftwf_plan_many_r2r(2, ia([H, W]), C, x.ctypes.data, None, C, 1, x.ctypes.data, None, C, 1, ia([5, 5]), 1 << 6)
        pi = fw.fftwf_plan_many_r2r(2, ia([H, W]), C, x.ctypes.data, None, C, 1, x.ctypes.data, None, C, 1, ia([4, 4]), 1 << 6)
        if not pf or not pi:
            return None
        reps, t0 = 0, time.perf_counter()
        while True:
            fw.fftwf_execute(pf); fw.fftwf_execute(pi)
            x *= np.float32(1.0 / (4.0 * W * H))
            reps += 1
            el = time.perf_counter() - t0
            if el > max_seconds / 2 or reps >= 32:
                break
        out[label] = (round(reps * H * W / 1e6 / el, 3), nthr, reps)
        fw.fftwf_destroy_plan(pf); fw.fftwf_destroy_plan(pi)
    if not out:
        return None
    best = out.get("all", out["1"])
    ver = C.c_char_p.in_dll(fw, "fftwf_version").value.decode()
    return {"value": best[0], "unit": "Mpixels/s", "cores": best[1], "kind": "fftw", "single_thread_value": out["1"][0],
            "sample": f"{best[2]} roundtrip(s) of one 3840x2160x3 f32 frame, {ver}, fftwf_plan_many_r2r(FFTW_ESTIMATE), {best[1]} thread(s)"}


def forward_check(torch, fwd, frame):
    """A check a no-op cannot pass (the drift of in-place roundtrips is zero for a kernel that does nothing): ONE forward transform of a
    copy of the frame against two properties computed from the INPUT by torch alone, in double --
    Y[0,0,c] = 4 sum(x[:,:,c]) (REDFT10 x REDFT10, FFTW's unnormalised definition) and the orthogonality relation
    sum' Y^2 = (2h)(2w) sum x^2 (sum' halves index 0 of each axis).  Returns the two relative errors."""
    x = frame.to(torch.float64)
    y = frame.clone()
    fwd.execute(y.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    y = y.to(torch.float64)
    dc_ref = 4.0 * x.sum(dim=(0, 1))
    dc_err = float(((y[0, 0] - dc_ref).abs() / dc_ref.abs().clamp_min(1e-30)).max())
    wy = torch.ones(H, dtype=torch.float64, device=frame.device); wy[0] = 0.5
    wx = torch.ones(W, dtype=torch.float64, device=frame.device); wx[0] = 0.5
    e_y = float((y * y * wy[:, None, None] * wx[None, :, None]).sum())
    e_x = float((x * x).sum()) * (2.0 * H) * (2.0 * W)
    return dc_err, abs(e_y - e_x) / e_x


def fftw_abi_end_to_end(reps=8):
    """SURVEY.md 8d "report separately the FFTW-ABI end-to-end time including H2D/D2H": ONE fftwf_execute of the 4K frame through
    include/fftw3.h on a pinned host buffer (fftwf_alloc_real) = H2D + two axis passes + D2H, synchronous on return.  Never `value`."""
    import ctypes as C_
    import numpy as np
    from dspfun_amd import _lib
    lib = C_.CDLL(_lib.LIB_PATH)
    ip = C_.POINTER(C_.c_int)
    lib.fftwf_alloc_real.restype = C_.c_void_p; lib.fftwf_alloc_real.argtypes = [C_.c_size_t]
    lib.fftwf_free.argtypes = [C_.c_void_p]
    lib.fftwf_plan_many_r2r.restype = C_.c_void_p
    lib.fftwf_plan_many_r2r.argtypes = [C_.c_int, ip, C_.c_int, C_.c_void_p, ip, C_.c_int, C_.c_int, C_.c_void_p, ip, C_.c_int, C_.c_int, ip, C_.c_uint]
    lib.fftwf_execute.argtypes = [C_.c_void_p]; lib.fftwf_destroy_plan.argtypes = [C_.c_void_p]
    n = H * W * C
    p = lib.fftwf_alloc_real(n)
    if not p:
        return {"error": "fftwf_alloc_real failed"}
    a = np.ctypeslib.as_array((C_.c_float * n).from_address(p))
    a[:] = np.random.default_rng(SEED).random(n, dtype=np.float32)
    ia = lambda v: (C_.c_int * len(v))(*v)
    fwd = lib.fftwf_plan_many_r2r(2, ia([H, W]), C, p, None, C, 1, p, None, C, 1, ia([5, 5]), 1 << 6)       # spec/spec.c:63
    inv = lib.fftwf_plan_many_r2r(2, ia([H, W]), C, p, None, C, 1, p, None, C, 1, ia([4, 4]), 1 << 6)       # spec/ispec.c:165
    for _ in range(2):
        lib.fftwf_execute(fwd); lib.fftwf_execute(inv); a *= np.float32(1.0 / (4.0 * W * H))
    t = 0.0
    for _ in range(reps):
        t0 = time.perf_counter(); lib.fftwf_execute(fwd); t += time.perf_counter() - t0
        lib.fftwf_execute(inv); a *= np.float32(1.0 / (4.0 * W * H))
    dt = t / reps
    lib.fftwf_destroy_plan(fwd); lib.fftwf_destroy_plan(inv)
    # what the host link itself does on THIS box, measured on the same pinned buffer: one plain copy each way (best of 5).  An execute moves the
    # frame up, transforms it, moves it down -- the download needs every uploaded byte, so the two transfers of one transform cannot overlap and
    # their sum is the floor of `ms`.  (Round 4 divided the bytes of ONE direction by the time of BOTH and called it the rate each way: half the truth.)
    import torch
    dv = torch.empty(n, dtype=torch.float32, device="cuda")
    hv = torch.frombuffer((C_.c_float * n).from_address(p), dtype=torch.float32)
    up, down = [], []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dv.copy_(hv, non_blocking=True); torch.cuda.synchronize(); up.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); hv.copy_(dv, non_blocking=True); torch.cuda.synchronize(); down.append(time.perf_counter() - t0)
    del hv
    lib.fftwf_free(p)
    floor = min(up) + min(down)
    return {"what": "one fftwf_execute (REDFT10 x REDFT10, in place) of a 3840x2160x3 f32 frame in pinned host memory: H2D + 2 axis passes + D2H, synchronous",
            "ms": round(dt * 1e3, 3), "Mpixels_per_s_one_direction": round(H * W / dt / 1e6, 1), "bytes_each_way": n * 4,
            "host_GBps_over_both_transfers": round(2 * n * 4 / dt / 1e9, 1),
            "link_GBps": {"h2d": round(n * 4 / min(up) / 1e9, 1), "d2h": round(n * 4 / min(down) / 1e9, 1), "what": "one hipMemcpy of the same pinned buffer each way, best of 5, this box"},
            "link_floor_ms": round(floor * 1e3, 3), "frac_of_link_floor": round(floor / dt, 3),
            "link_floor_note": "upload + download of ONE transform cannot overlap (the download needs every uploaded byte): their sum is the floor",
            "executes_timed": reps}


def self_launch(ngpus):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a child (one rank per GPU, RCCL rendezvous on 127.0.0.1) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--frames", type=int, default=4, help="independent frames per GPU per step")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("BENCH_STREAMS", "2")), help="HIP streams the independent frames are spread over")
    ap.add_argument("--realign", type=int, default=8, help="with --schedule realign: re-join the streams every N steps")
    ap.add_argument("--schedule", choices=["free", "aligned", "realign"], default="realign", help="how two streams interleave their frames (see step())")
    ap.add_argument("--event-every", type=int, default=8, help="bracket the passes of one frame with events every N-th step (events between "
                    "dependent launches cost throughput: every step -3 %%, every launch -7 %%)")
    ap.add_argument("--inverse-order", choices=["columns-first", "rows-first"], default="columns-first", help="axis order of the REDFT01 plan")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-stream", action="store_true", help="skip the single_stream_value leg (the same frames on ONE stream, after the headline)")
    ap.add_argument("--no-fftw-abi", action="store_true", help="skip the fftw_abi_end_to_end object (one fftwf_execute on pinned host memory)")
    ap.add_argument("--no-motion", action="store_true", help="skip the motion_c5 object (BASELINE configs[4]: per-frame blocks and the RCCL slab volume)")
    ap.add_argument("--extras-timeout", type=float, default=300.0, help="N > 1 only: seconds the motion_c5 / scan_c4 objects may take after the headline before the line is printed without them")
    ap.add_argument("--no-scan", action="store_true", help="skip the scan_c4 object (BASELINE configs[3]: channel-sharded progressive reconstruct of an 8K frame)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called directly as `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process of a parent
        # that has not imported torch or touched the GPU (a process that has initialised the GPU must never exec)
        sys.exit(self_launch(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    dist = None
    # DSPFFT_BENCH_FORCE_DIST=1 runs the RCCL path (init, barriers, max-over-ranks) with a single rank: the one way to exercise it on a 1-GPU box
    if world > 1 or (os.environ.get("DSPFFT_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    # how many ranks RCCL itself joined (an all-reduce of ones): what the collectives below really span, whatever WORLD_SIZE says
    ranks_seen = None
    if dist is not None:
        one = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    from dspfun_amd import Plan, REDFT10, REDFT01
    fwd = Plan.image(H, W, C, REDFT10)
    # the inverse plan runs its column pass first (dspfft_plan_many_r2r_ordered: the planner's choice of axis order, same results): a
    # frame's four launches are then ROW, COL | COL, ROW and each stream changes workgroup shape twice per frame instead of four
    # times, so the two streams overlap same-shaped kernels more of the time (+1 %, tools/cbench.c)
    inv = Plan.many_r2r([H, W], [REDFT01] * 2, howmany=C, istride=C, idist=1, ostride=C, odist=1,
                        first_axis_first=(args.inverse_order == "columns-first")).set_scale(1.0 / (4.0 * W * H))
    frames = synth_frames(torch, args.frames, dev)
    ref0 = frames[0].clone()
    fwd_dc_err, fwd_energy_err = forward_check(torch, fwd, frames[0])      # what a no-op cannot pass (the in-place drift below can)
    torch.cuda.synchronize()          # the frames are used on other streams from here on
    stream = torch.cuda.current_stream().cuda_stream
    ptrs = [frames[f].data_ptr() for f in range(args.frames)]

    # independent frames may run on separate HIP streams: kernels of different frames (and different
    # pass shapes) then share the CUs, so one frame's memory-bound phases overlap another's butterflies
    nstreams = max(1, min(args.streams, args.frames))
    # the streams are the library's own (dspfft_stream_create = hipStreamCreateWithFlags(hipStreamNonBlocking)): two streams from
    # torch's pool were seen sharing a hardware queue, which serialises the frames they carry (tools/py_enqueue_probe.py: 48K vs 56K)
    from dspfun_amd.engine import Stream
    side = [Stream() for _ in range(nstreams)] if nstreams > 1 else []
    handles = [s_.handle for s_ in side] if side else [stream]

    from dspfun_amd.engine import Batch, Events
    npass = fwd.num_passes + inv.num_passes
    # One step = ONE call into the library (dspfft_execute_many: frame f -> forward plan, inverse plan, on stream f mod nstreams), so
    # the Python binding costs one ctypes call per step instead of one per pass (measured: 16 ctypes calls + 8 torch event records
    # per 290-us step left the GPU waiting for the host -- the same launches from C ran 3-5 % faster, tools/sbench.hip).
    # Events of the library bracket the passes of ONE frame per step inside the timed region (on the stream the kernels are launched
    # on), rotating through the frames, so the in-region average per kernel samples every position in the step -- the population
    # rocprofv3 averages -- at a quarter of the event traffic.
    batch = Batch([(pl_, p, None, handles[i % len(handles)]) for i, p in enumerate(ptrs) for pl_ in (fwd, inv)])
    every = max(1, args.event_every)
    nper = min(2, len(ptrs)) if nstreams >= 2 else 1          # frames bracketed per timed step: one per stream
    if len(ptrs) % nper:
        print(f"bench.py: --frames {len(ptrs)} must be a multiple of {nper} (the frames bracketed per timed step)", file=sys.stderr)
        sys.exit(2)
    timed_steps = [k for k in range(args.steps) if k % every == 0]
    events = Events(2 * npass * nper * len(timed_steps))

    # Schedules of the frames of a step on two streams (frame f on stream f mod 2).  Left alone ("free") the streams keep whatever
    # relative phase they happen to have: 58-59K Mpix/s while kernels of the same pass shape overlap (tools/sbench.hip, both boxes
    # seen), 45K when a stream's row passes run beside the other's column passes (mixed workgroup shapes leave LDS unused; forced:
    # "pipe" 35-39K) -- and anything that delays one stream (the timing events below, a colder frame) moves the phase.  Re-joining
    # the streams at EVERY step ("aligned", round 1's default) is steady but idles the faster stream at each join (55-56K).
    # Default: re-join every `--realign` steps (8): the phase cannot wander, the joins are rare.  The timing events bracket one
    # frame on EACH stream in the same step, so they delay both streams alike.
    # Round 3: the step loop itself is the library's (dspfft_execute_many_repeat: a clip's frame loop, K steps in ONE call), so no
    # host-language code runs between two steps; the re-joins are the library's too.
    rejoin = 0
    schedule = "single stream" if nstreams == 1 else "free"
    if nstreams >= 2 and args.schedule == "aligned":
        rejoin, schedule = 1, "step-aligned"
    elif nstreams >= 2 and args.schedule == "realign":
        rejoin = max(1, args.realign)
        schedule = f"free, streams re-joined every {rejoin} steps"

    bar = {"ms": [0.0, 0.0, 0.0]}

    def barrier():
        a0 = time.perf_counter()
        torch.cuda.synchronize()
        a1 = time.perf_counter()
        if dist is not None:
            dist.barrier()
        a2 = time.perf_counter()
        torch.cuda.synchronize()
        bar["ms"] = [round((a1 - a0) * 1e3, 3), round((a2 - a1) * 1e3, 3), round((time.perf_counter() - a2) * 1e3, 3)]

    # clocks and caches settle over a few tens of milliseconds: whatever --warmup says, run at least that much untimed work
    # first (a short --steps run is otherwise timed on a GPU that is still ramping up)
    # The FIRST dist.barrier() of a process costs the work behind it about 1.7 ms, once (tools/dist_gap_probe.py: 20 steps take 13.2 ms
    # behind the first barrier, 11.6 ms behind any later one -- RCCL still setting itself up in the background): take it here, before
    # the warm-up, so that the barrier which opens the timed region is not the first.
    barrier()
    # (700 steps = 0.4 s: on a box that has just been handed out the first process measured 55.1-56.1K Mpix/s behind 60 untimed steps and
    # 57.4-57.8K behind 700, `--steps 20 --warmup 5` both times; a second process on the same box 56.4-56.6K against 57.6-57.9K)
    preroll = int(os.environ.get("DSPFFT_BENCH_PREROLL", "700"))
    if args.warmup < preroll:
        batch.run_repeat(preroll - args.warmup, rejoin)
    batch.run_repeat(args.warmup, rejoin)
    barrier()
    t0 = time.perf_counter()
    batch.run_repeat(args.steps, rejoin, every, 2 * nper, events)      # EXACTLY --steps steps
    enqueue_s = time.perf_counter() - t0          # host time to enqueue all steps (must stay well below the GPU's time)
    barrier()
    elapsed = time.perf_counter() - t0
    closing = list(bar["ms"])                     # [synchronize, dist.barrier, synchronize] of the closing bracket, ms (inside the timed region)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the same batch on ONE stream (each frame's four launches alone on the chip), untimed-warm, then K steps between synchronisations:
    # what the two-stream schedule buys is value / single_stream_value
    single_stream_value = None
    if nstreams > 1 and not args.no_single_stream:
        b1 = Batch([(pl_, p, None, handles[0]) for p in ptrs for pl_ in (fwd, inv)])
        n1 = max(args.steps, 20)
        b1.run_repeat(max(args.warmup, 10), 0)
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        b1.run_repeat(n1, 0)
        torch.cuda.synchronize()
        single_stream_value = round(n1 * args.frames * H * W / 1e6 / (time.perf_counter() - s0), 2)
    # SURVEY 8d's timing protocol beside the batch throughput: ONE frame's forward + inverse pair alone on the chip, hipEvents around the pair on
    # the stream it is launched on, 10 pairs untimed, then 60: median and minimum (what a caller that transforms frame after frame on one
    # stream sees per frame; `value` above is four frames on two streams sharing the CUs)
    lat = []
    for i_ in range(70):
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        fwd.execute(ptrs[0], stream=stream); inv.execute(ptrs[0], stream=stream)
        b_.record()
        lat.append((a_, b_))
    torch.cuda.synchronize()
    lat = sorted(a_.elapsed_time(b_) for a_, b_ in lat[10:])
    frame_latency = {"median": round(lat[len(lat) // 2], 5), "min": round(lat[0], 5), "pairs_timed": len(lat), "pairs_untimed": 10,
                     "what": "one 3840x2160x3 frame, REDFT10^2 then REDFT01^2 in place (four launches), alone on one stream, HIP events around the pair",
                     "Mpixels_per_s_at_median": round(H * W / lat[len(lat) // 2] / 1e3, 1)}
    untimed_steps = max(preroll, args.warmup)
    nt = len(timed_steps) * nper
    in_region_ms = [sum(events.elapsed_ms(2 * npass * t + 2 * j, 2 * npass * t + 2 * j + 1) for t in range(nt)) / nt for j in range(npass)]

    # sanity of what was timed: after (warmup+steps) consecutive in-place roundtrips the frame is still the input
    # (single-roundtrip accuracy is what tests/test_gpu_parity.py pins: <= 5e-6)
    drift = float((frames[0] - ref0).abs().max())
    global DRIFT_BOUND
    roundtrips_of_frame0 = untimed_steps + args.steps + (max(args.steps, 20) + max(args.warmup, 10) if single_stream_value is not None else 0) + 70
    DRIFT_BOUND = max(DRIFT_BOUND, roundtrips_of_frame0 * DRIFT_PER_ROUNDTRIP)     # measured: 8.4e-7 per roundtrip

    # roofline object (rank 0): dominant kernel = longest average in-region launch
    roof = None
    if rank == 0:
        names, iso = [], []
        reps = 22
        for plan, tag in ((fwd, "redft10"), (inv, "redft01")):
            desc = [ln for ln in plan.describe().splitlines() if ln.startswith("axis")]
            for i in range(plan.num_passes):
                e2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
                for a, b in e2:
                    a.record()
                    plan.execute_pass(i, ptrs[0], stream=stream)
                    b.record()
                torch.cuda.synchronize()
                ms = [a.elapsed_time(b) for a, b in e2][2:]                  # first two launches warm the caches
                iso.append(sum(ms) / len(ms))
                names.append(f"{tag} {desc[i].split(':')[1].split()[0]} pass ({desc[i].strip()})")
        frames[0].copy_(ref0)
        k = max(range(npass), key=lambda i: in_region_ms[i])
        # algorithmic bytes of ONE launch: the 48 B/pixel roundtrip figure is 4 axis passes of 12 B/pixel each
        alg = H * W * ALG_BYTES_PER_PIXEL / 4.0
        achieved = alg / (in_region_ms[k] * 1e-3)
        # PMC traffic is not measured by this run (rocprofv3 --pmc needs its own passes: tools/pmc_traffic.sh); it is looked up
        # in profiles/traffic.json by the KIND of the dominant pass and dropped when that file does not describe the kernels this
        # plan runs (its "plan" field must equal the forward plan's description)
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = ("row" if " ROW" in names[k] else "col") + "_" + names[k].split()[0]
                if tj.get("plan") == [ln.strip() for ln in fwd.describe().splitlines() if ln.startswith("axis")]:
                    traffic = tj.get("hbm_bytes_per_launch", {}).get(key)
                    traffic_source = f"profiles/traffic.json ({tj.get('round', '?')}), key {key}"
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK, 4), "traffic": traffic,
                "traffic_source": (f"PROFILE LOOKUP, not measured by this run: {traffic_source}" if traffic is not None else None),
                # what the chip physically moved per launch (counter bytes: the frame once in, once out) over the launch's duration: the kernel's
                # memory rate, against the 8 TB/s peak; `frac` charges the launch only its quarter of the 48 B/pixel roundtrip figure
                "physical_frac": (round(traffic / (in_region_ms[k] * 1e-3) / HBM_PEAK, 4) if traffic is not None else None),
                "kernel": names[k], "kernel_ms": round(in_region_ms[k], 5),
                "all_kernels_ms": {names[i]: round(in_region_ms[i], 5) for i in range(npass)},
                "algorithmic_bytes_per_launch": alg, "launches_timed_in_region_per_kernel": nt,
                "note": ("durations are HIP-event averages over the timed region; with hip_streams > 1 the kernels of two frames "
                         "share the CUs, so a launch lasts longer than when it runs alone (isolated_*)") if nstreams > 1 else
                        "durations are HIP-event averages over the timed region",
                "isolated_kernel_ms": {names[i]: round(iso[i], 5) for i in range(npass)},
                "isolated_frac_dominant": round(alg / (max(iso) * 1e-3) / HBM_PEAK, 4)}

    # BASELINE configs[4] beside the headline: motion's per-frame blocks (frame-sharded, no collective; strong and
    # weak) and its one-3-D-block mode through SlabDCT3D -- the one path that exercises RCCL all-to-all -- so that a SCALE record at
    # N = 2, 4, 8 carries the curve north_star asks for.  Every rank takes part; its own barriers and max-over-ranks timing.
    motion = None
    del frames, ref0
    # On N > 1 ranks the extras below run collectives (RCCL all-to-all, all_gather) that no box of this round could exercise: if one of them
    # hangs, the headline measured above must still be printed.  A watchdog per rank: past --extras-timeout seconds rank 0 prints the line
    # with what has finished (the unfinished object says so) and every rank leaves without waiting for the others.
    extras = {"motion": None, "scan": None, "done": False}
    finished = threading.Event()

    def make_line(motion, scan):
        pixels = args.steps * args.frames * world * H * W
        value = pixels / 1e6 / elapsed
        line = {
            "metric": "Mpixels/s DCT-II+III roundtrip 3840x2160x3", "value": round(value, 2), "unit": "Mpixels/s",
            "n_gpus": world, "ranks_seen_by_rccl": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (splitmix64 uniform [0,1), SURVEY.md 8d seed 0xD5F0002)",
            "config": {"workload": "spec + ispec roundtrip on 3840x2160 RGB float32 (BASELINE configs[1])",
                       "frames_per_gpu_per_step": args.frames, "hip_streams": nstreams, "stream_schedule": schedule,
                       "batch": f"batch of {args.frames} independent frames on {nstreams} HIP stream(s) per step: `value` is batch throughput, not one frame's latency",
                       "preroll_steps": untimed_steps - args.warmup if untimed_steps > args.warmup else 0, "untimed_steps_total": untimed_steps, "step_loop": "dspfft_execute_many_repeat (one library call for all steps)", "inverse_plan_order": args.inverse_order, "layout": "interleaved HWC, in place, device-resident",
                       "parallelism": f"frame-sharded x{world}, no collective"},
            "roundtrip_frac_of_hbm_roofline": round(value * 1e6 * ALG_BYTES_PER_PIXEL / (HBM_PEAK * world), 4),      # per GPU
            # north_star asks for >= 70 % of the HBM roofline on this roundtrip.  A plane is 200 times what a CU holds and the row -> column exchange
            # crosses XCDs, so each direction is two passes over memory: four passes per roundtrip at the no-arithmetic floors of their access shapes
            # (37.5 + 34.5 + 34.5 + 29.5 us, profiles/r02_membench2.csv) are 0.366 of the 48 B/pixel roofline -- the declared cap of this design
            # (the TARGET is north_star's 0.70 and is not met; `builders_model_of_this_design` is the builder's own estimate of what a two-pass-per-direction
            # design can reach, reported for context: not a goal post -- ADVICE r05)
            "target": {"roundtrip_frac_of_hbm_roofline": 0.70, "met": bool(value * 1e6 * ALG_BYTES_PER_PIXEL / (HBM_PEAK * world) >= 0.70),
                       "builders_model_of_this_design": {"cap": 0.366, "frac_of_it": round(value * 1e6 * ALG_BYTES_PER_PIXEL / (HBM_PEAK * world) / 0.366, 3),
                                                         "why": "two memory passes per direction are forced (no plane fits on chip); four passes at the copy floors of their access shapes = 0.366"}},
            "single_stream_value": single_stream_value,
            "frame_latency_ms": frame_latency,
            "forward_check": {"dc_rel_err": fwd_dc_err, "energy_rel_err": fwd_energy_err,
                              "what": "one REDFT10 x REDFT10 of frame 0 before the timed region: Y[0,0] = 4 sum(x) and sum' Y^2 = 4hw sum x^2 (a no-op fails both)"},
            "max_abs_drift_after_all_roundtrips": drift, "roundtrips_of_frame0": roundtrips_of_frame0, "host_enqueue_ms_per_step": round(enqueue_s / args.steps * 1e3, 5),
            "closing_bracket_ms": {"synchronize": closing[0], "barrier": closing[1], "synchronize_after": closing[2]},
            "roofline": roof,
        }
        if motion is not None:
            line["motion_c5"] = motion
        if scan is not None:
            line["scan_c4"] = scan
        # a run whose frames no longer equal the input after all the roundtrips did not time the transform: no headline number
        if not (drift <= DRIFT_BOUND):
            line["value"] = None
            line["error"] = f"max_abs_drift_after_all_roundtrips {drift} exceeds {DRIFT_BOUND}"
        if not (fwd_dc_err <= 1e-4 and fwd_energy_err <= 1e-4):
            line["value"] = None
            line["error"] = f"forward_check failed: the forward plan did not transform the frame (dc {fwd_dc_err}, energy {fwd_energy_err})"
        return line

    def watchdog():
        if finished.wait(args.extras_timeout):
            return
        late = {"error": f"not finished {args.extras_timeout} s after the headline (a collective that hangs?); the headline above stands"}
        if rank == 0:
            print(json.dumps(make_line(extras["motion"] if extras["motion"] is not None or args.no_motion else late,
                                       extras["scan"] if extras["scan"] is not None or args.no_scan else late)), flush=True)
        os._exit(0 if drift <= DRIFT_BOUND and fwd_dc_err <= 1e-4 and fwd_energy_err <= 1e-4 else 1)

    if dist is not None and not (args.no_motion and args.no_scan):
        threading.Thread(target=watchdog, daemon=True).start()
    if not args.no_motion:
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from bench_motion import motion_c5
        try:
            motion = motion_c5(torch, dist, dev, rank, world)
        except Exception as e:          # the headline line must still be printed
            motion = {"error": f"{type(e).__name__}: {e}"}
        extras["motion"] = motion

    scan = None
    if not args.no_scan:
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from bench_scan_c4 import scan_c4
        try:
            scan = scan_c4(torch, dist, dev, rank, world)
        except Exception as e:
            scan = {"error": f"{type(e).__name__}: {e}"}
        extras["scan"] = scan
    finished.set()

    if rank == 0:
        line = make_line(motion, scan)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
            fw_ = fftw_cpu_baseline()
            # the oracle's port stays THE cpu_baseline object (kind "port"); a real FFTW, when the box has one, is reported beside it
            line["cpu_baseline_fftw"] = fw_ if fw_ is not None else "libfftw3f.so.3 not present on this box"
            sp_ = scipy_cpu_baseline()
            line["cpu_baseline_scipy"] = sp_ if sp_ is not None else "scipy does not import on this box"
        if not args.no_fftw_abi and world == 1:
            try:
                line["fftw_abi_end_to_end"] = fftw_abi_end_to_end()
            except Exception as e:
                line["fftw_abi_end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
        bad = line["value"] is None
        print(json.dumps(line), flush=True)
        status = 1 if bad else 0
    else:
        status = 0
    if dist is not None:
        st_ = torch.tensor([status], dtype=torch.int32, device=dev)
        dist.broadcast(st_, src=0)
        status = int(st_.item())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if status:
        print("bench.py: the frames drifted from the input; see the JSON line", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
