"""CPU: the multi-process path (dspfun_amd/dist.py) with world_size 2 over gloo.  The local transforms
run on the test-only emulation backend (there is no GPU here); what is under test is the sharding
and the all-to-all re-slabbing, against the single-process oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, d, h, w, chunks, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as ol
        from emul_lib import emul
        from dspfun_amd.dist import SlabDCT3D, block_range
        vol = ol.synth_u8(0xD5F0005, d * h * w).astype(np.float32).reshape(d, h, w)
        lo, hi = block_range(d, rank, world)
        ylo, yhi = block_range(h, rank, world)
        eng = SlabDCT3D(d, h, w, lib=emul(), chunks=chunks)
        assert (eng.f_lo, eng.f_hi, eng.y_lo, eng.y_hi) == (lo, hi, ylo, yhi)
        mine = torch.from_numpy(vol[lo:hi].copy())
        c = eng.forward(mine)
        assert tuple(c.shape) == (d, yhi - ylo, w)
        # reference: REDFT10^3 with motion's uniform scaling, rows of my slab
        ref = ol.r2r_many(vol.astype(np.float64), [d, h, w], [ol.REDFT10] * 3, impl="port")
        ol.lib().oracle_motion_uniform_f64(ref.ctypes.data, d, h, w, h, w, 1)
        ref = ref.reshape(d, h, w)
        err_f = np.abs(c.numpy() - ref[:, ylo:yhi, :]).max() / np.abs(ref).max() if yhi > ylo else 0.0
        keep = c.clone()
        back = eng.inverse(c)
        assert torch.equal(c, keep)                      # the coefficients are not consumed
        err_b = np.abs(back.numpy() - vol[lo:hi]).max() if hi > lo else 0.0
        # a second roundtrip through the same (reused) exchange buffers
        back2 = eng.inverse(eng.forward(torch.from_numpy(vol[lo:hi].copy())))
        err_b = max(err_b, float(np.abs(back2.numpy() - vol[lo:hi]).max()) if hi > lo else 0.0)
        # the exchanges alone (what tools/bench_motion.py times for `exchange_ms` / `xgmi_frac`): every rank sends (G - 1) / G of its buffer
        sent = eng.exchange_alone(mine)
        assert sent == eng.P * eng.G * eng.dlp * eng.ch * eng.w * 4 * (world - 1) // world
        q.put((rank, float(err_f), float(err_b)))
    finally:
        dist.destroy_process_group()


# (d, h, w, world, chunks): even splits; h % G != 0 and d % G != 0 (BASELINE config 5's chroma planes: 540 rows on 8 ranks);
# a rank with no rows at all; one piece and more pieces than rows
# world 4 and 8 (the node sizes the SCALE record uses) with h % G != 0 and d % G != 0: 16 x 27 x 8 on 4 and on 8 ranks, 10 frames on 8 ranks
@pytest.mark.parametrize("d,h,w,world,chunks", [(8, 12, 10, 2, 1), (16, 30, 24, 2, 4), (6, 27, 10, 2, 3), (7, 10, 12, 3, 2), (4, 5, 8, 3, 8), (9, 2, 8, 3, 1),
                                                (16, 27, 8, 4, 2), (16, 27, 8, 8, 3), (10, 12, 16, 8, 1),
                                                # BASELINE config 5's real chroma row count on 8 ranks: 540 rows = 7 blocks of 68 + one of 64, four pieces of 17
                                                # rows (the short block's last piece has 13) -- width and depth reduced so that the emulation finishes in seconds
                                                (8, 540, 8, 8, 4)])
def test_slab_dct3d_gloo(d, h, w, world, chunks):
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emul")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(HERE), "oracle"), "liboracle.so"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, d, h, w, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err_f, err_b in res:
        assert err_f < 1e-5, (rank, err_f)
        assert err_b < 2e-3, (rank, err_b)       # pixel domain 0..255: far below half an LSB


def test_block_range_is_equal_blocks():
    from dspfun_amd.dist import block_range
    for n, world in ((540, 8), (256, 8), (10, 8), (1, 3), (7, 2)):
        spans = [block_range(n, r, world) for r in range(world)]
        per = -(-n // world)
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert all(b - a == per for a, b in spans if b < n)
    assert block_range(540, 7, 8) == (476, 540) and block_range(10, 6, 8) == (10, 10)


def test_shard_range_covers_everything():
    from dspfun_amd.dist import shard_range
    for n in (1, 7, 8, 256, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _scan_worker(rank, world, port, h, w, step, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as ol
        from emul_lib import emul
        from dspfun_amd.dist import ChannelShardedScan
        x = ol.synth_f32(0xD5F0004, h * w * 3).reshape(h, w, 3)
        eng = ChannelShardedScan(torch.from_numpy(x.copy()), step, lib=emul())
        # reference after 2 frames and at the end (f64 restatement on the interleaved image)
        c64 = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10))
        ol.lib().oracle_scan_normalise_f64(c64.ctypes.data, w, h, 3)
        ref = np.ascontiguousarray(np.broadcast_to(c64[0, 0], (h, w, 3)).copy())
        zz = ol.zigzag_order(w, h)
        errs = []
        k = 0
        while eng.next_frame():
            lin = np.ascontiguousarray(zz[k * step:(k + 1) * step])
            ol.lib().oracle_scan_frame_f64(w, h, 3, c64.ctypes.data, lin.ctypes.data, lin.size, ref.ctypes.data)
            k += 1
            if k in (2, eng.nframes):
                errs.append(float(np.abs(eng.gather().numpy() - ref).max()))
        errs.append(float(np.abs(eng.gather().numpy() - x).max()))
        q.put((rank, len(eng.mine), errs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", ["auto", "planar"])
def test_channel_sharded_scan_one_rank_layouts(layout):
    """no process group: the only rank owns the three planes and keeps the image interleaved (one execution per frame); "planar" forces the
    per-plane form the larger worlds use -- both against the f64 restatement of scan.c:421-459"""
    import oracle_lib as ol
    from emul_lib import emul
    from dspfun_amd.dist import ChannelShardedScan
    h, w, step = 12, 20, 37
    x = ol.synth_f32(0xD5F0004, h * w * 3).reshape(h, w, 3)
    eng = ChannelShardedScan(torch.from_numpy(x.copy()), step, lib=emul(), layout=layout)
    assert eng.interleaved == (layout == "auto") and eng.mine == [0, 1, 2]
    before = eng.gather().numpy().copy()
    eng.warm()                                              # a frame id nobody owns: the sums stay bit for bit
    assert np.array_equal(before, eng.gather().numpy())
    c64 = np.ascontiguousarray(ol.dct2d_interleaved(x.astype(np.float64), ol.REDFT10))
    ol.lib().oracle_scan_normalise_f64(c64.ctypes.data, w, h, 3)
    ref = np.ascontiguousarray(np.broadcast_to(c64[0, 0], (h, w, 3)).copy())
    zz = ol.zigzag_order(w, h)
    k = 0
    while eng.next_frame():
        lin = np.ascontiguousarray(zz[k * step:(k + 1) * step])
        ol.lib().oracle_scan_frame_f64(w, h, 3, c64.ctypes.data, lin.ctypes.data, lin.size, ref.ctypes.data)
        k += 1
        assert float(np.abs(eng.gather().numpy() - ref).max()) < 5e-6, k
    assert k == eng.nframes and float(np.abs(eng.gather().numpy() - x).max()) < 5e-6


@pytest.mark.parametrize("world", [2, 4])
def test_channel_sharded_scan(world):
    """BASELINE config 4 names 4 GPUs for 3 colour planes: with world 4 the fourth rank owns no plane (75 % ceiling, stated in DESIGN 6)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scan_worker, args=(r, world, port, 12, 20, 37, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(n for _, n, _ in res) == ([1, 2] if world == 2 else [0, 1, 1, 1])          # 3 planes over the ranks
    for rank, _, errs in res:
        assert max(errs) < 1e-5, (rank, errs)


def _scan_c4_worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emul_lib import emul
        from bench_scan_c4 import scan_c4
        q.put((rank, scan_c4(torch, dist, torch.device("cpu"), rank, world, size=(40, 24), step=97, lib=emul())))
    finally:
        dist.destroy_process_group()


def test_scan_c4_bench_path_on_four_ranks():
    """tools/bench_scan_c4.scan_c4 -- the function bench.py calls for its `scan_c4` object, barriers, all_reduce of the time and final gather
    included -- under four gloo ranks on a small frame with the emulation library: the driver's first `bench.py --gpus 4` is then not the first
    time this code runs with more than one rank"""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scan_c4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        o = res[rank]
        assert o["frames"] == -(-40 * 24 // 97) and o["planes_per_rank"] == [1, 1, 1, 0] and o["ranks_with_a_plane"] == 3
        assert o["scaling_efficiency_ceiling"] == 0.75 and o["max_abs_final_sum_minus_input"] < 5e-6 and o["ms_per_frame"] > 0
        assert o["ms_per_frame_second_scan"] > 0
        assert "planar" in o["layout"]
    assert len({res[r]["ms_per_frame"] for r in range(world)}) == 1          # the MAX over ranks, agreed by all_reduce


def _motion_c5_worker(rank, world, port, frames, planes_hw, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emul_lib import emul
        from bench_motion import motion_c5
        q.put((rank, motion_c5(torch, dist, torch.device("cpu"), rank, world, reps_frames=4, reps_volume=2, frames=frames, planes_hw=planes_hw, lib=emul())))
    finally:
        dist.destroy_process_group()


def test_motion_c5_bench_path_on_eight_ranks():
    """tools/bench_motion.motion_c5 -- the function bench.py calls for its `motion_c5` object: frames_bench strong and weak, volume_bench and
    SlabDCT3D.exchange_alone, with their barriers and the all_reduce of the time -- under eight gloo ranks with the emulation library on a clip
    whose frame count and row counts do not divide by eight (motion/motion.c:535-552,591,613-615): the driver's first `bench.py --gpus 8` is then
    not the first time this code runs with more than one rank.  The exchange figures are recomputed here from the object's own fields."""
    world, frames, planes_hw = 8, 20, ((20, 24), (10, 12), (10, 12))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_motion_c5_worker, args=(r, world, port, frames, planes_hw, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from dspfun_amd.dist import shard_range, block_range
    # what one rank sends to the others in the all-to-alls of one direction (SlabDCT3D._buf / exchange_alone), recomputed from the geometry
    sent = 0
    for (h, w) in planes_hw:
        dlp, hp = -(-frames // world), -(-h // world)
        ch = -(-hp // min(4, hp))
        pieces = -(-hp // ch)
        sent += pieces * world * dlp * ch * w * 4 * (world - 1) // world
    for rank in range(world):
        o = res[rank]
        s, wk, v = o["per_frame_strong"], o["per_frame_weak"], o["volume_3d"]
        lo, hi = shard_range(frames, rank, world)
        assert s["scaling"] == "strong" and s["frames_per_rank"] == hi - lo and s["ms_per_clip_round"] > 0 and 0 <= s["max_abs_u8_change"] < 256
        assert wk["scaling"] == "weak" and wk["frames_per_rank"] == frames and wk["ms_per_clip_round"] > 0
        assert "error" not in v, v
        flo, fhi = block_range(frames, rank, world)
        assert v["frames_per_rank"] == fhi - flo and v["ms_per_clip"] > 0 and v["exchanges_per_clip"] == 6
        assert v["max_abs_roundtrip_error_0_255"] < 1e-3
        assert v["exchange_ms"] is not None and v["exchange_ms"] > 0 and v["exchange_bytes_sent_per_rank"] == sent
        frac = sent / (v["exchange_ms"] * 1e-3) / (7 * 76.8e9)
        assert v["xgmi_frac"] is not None and abs(v["xgmi_frac"] - frac) <= 2e-3 * frac
        assert abs(v["xgmi_GBps_sent_per_rank"] - sent / (v["exchange_ms"] * 1e-3) / 1e9) <= 2e-3 * v["xgmi_GBps_sent_per_rank"]
    # the MAX over ranks, agreed by all_reduce
    for key in (("per_frame_strong", "ms_per_clip_round"), ("per_frame_weak", "ms_per_clip_round"), ("volume_3d", "ms_per_clip"), ("volume_3d", "exchange_ms")):
        assert len({res[r][key[0]][key[1]] for r in range(world)}) == 1, key
    assert sum(res[r]["per_frame_strong"]["frames_per_rank"] for r in range(world)) == frames


def test_x_pass_launches_per_piece():
    """one x-pass launch per piece for all full blocks (a third batch level of the row pass) + at most one for a short last block,
    whatever the number of ranks (round 3: one per rank)"""
    from dspfun_amd.dist import SlabDCT3D
    for d, h, G, chunks in ((256, 540, 8, 4), (256, 1080, 8, 4), (16, 27, 8, 3), (10, 12, 16, 1), (7, 10, 3, 2)):
        eng = SlabDCT3D.__new__(SlabDCT3D)
        eng.G, eng.h, eng.hp = G, h, -(-h // G)
        eng.ch = -(-eng.hp // max(1, min(chunks, eng.hp)))
        eng.P = -(-eng.hp // eng.ch)
        rows = 0
        for p in range(eng.P):
            g = eng._groups(p)
            assert len(g) <= 2
            covered = []
            for r0, nb, n in g:
                for r in range(r0, r0 + nb):
                    y0, y1 = eng._rows_of(r, p)
                    assert y1 - y0 == n and y0 == eng._rows_of(r0, p)[0] + (r - r0) * eng.hp
                    covered.append(r)
                    rows += n
            assert covered == [r for r in range(G) if eng._rows_of(r, p)[1] > eng._rows_of(r, p)[0]]
        assert rows == h


def test_row_pass_with_three_batch_levels_matches_the_oracle():
    """guru plan: transform along x, batch (rows, frames, blocks) with independent in / out strides -- ONE launch (no host loop)"""
    sys.path.insert(0, HERE)
    import ctypes as C
    import oracle_lib as ol
    from emul_lib import emul
    from dspfun_amd.engine import Plan, REDFT10
    L = emul()
    w, nrows, dl, nb, h, hp, ch, dlp = 16, 3, 2, 3, 12, 4, 3, 2
    x = ol.synth_f32(3, dl * h * w).reshape(dl, h, w)
    out = np.full((nb, dlp, ch, w), np.float32(-7))
    p = Plan.guru([(w, 1, 1)], [(nrows, w, w), (dl, h * w, ch * w), (nb, hp * w, dlp * ch * w)], [REDFT10], lib=L)
    assert "hostloop" not in p.describe()
    src = np.ascontiguousarray(x)
    p.execute(src.ctypes.data, out.ctypes.data)
    for b in range(nb):
        for f in range(dl):
            for y in range(nrows):
                ref = ol.r2r_many(x[f, b * hp + y].astype(np.float64), [w], [ol.REDFT10])
                assert np.abs(out[b, f, y] - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.all(out[:, :, nrows:, :] == -7) if ch > nrows else True
