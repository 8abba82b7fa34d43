"""Multi-GPU sharding of the hot path, one process per GPU over torch.distributed
(backend "nccl" == RCCL over xGMI on the MI355X node; "gloo" in the CPU tests).

Two regimes:

* independent units -- frames of motion's default `-b 0x0x1` (motion/motion.c:174,613-615), channel
  planes or whole images: `shard_range()` gives each rank a contiguous range; NO collective on the
  data path (bench.py uses this: weak scaling).

* one 3-D block spanning the whole clip (`-b 0x0x0`, motion/motion.c:535-552: REDFT10^3 / REDFT01^3 on a
  {d,h,w} block): `SlabDCT3D`.  Each rank owns a block of consecutive frames.  Forward = local y and x
  passes -> all-to-all that re-slabs the volume by rows (each rank then owns a block of rows of ALL
  frames) -> local z pass.  The coefficients stay in the z-local layout, where motion's filters
  (elementwise, motion.c:650-744) can run; inverse = z pass -> all-to-all back -> x and y passes.
  The x pass writes / reads the exchange buffers itself (no pack or unpack sweep), the exchange is cut into
  pieces that overlap the neighbouring pieces' passes, and d, h need not divide by the number of ranks.
  No all-reduce anywhere; xGMI is point-to-point and an all-to-all uses all 7 links at once.
"""
import math

import torch
import torch.distributed as dist

from .engine import Plan, REDFT01, REDFT10


def shard_range(n, rank, world):
    """contiguous [lo, hi) of n independent units for `rank` (sizes differ by at most one)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def block_range(n, rank, world):
    """[lo, hi) of rank's block when n units are dealt in EQUAL blocks of ceil(n / world) (the last blocks may be short or
    empty): the layout SlabDCT3D uses for frames and for rows, because equal blocks make every exchange buffer an affine
    array (no gaps to pack around) whatever n % world is"""
    per = -(-n // world)
    return min(n, rank * per), min(n, (rank + 1) * per)


class SlabDCT3D:
    """Distributed 3-D DCT-II / DCT-III of a {d,h,w} f32 volume (motion -b 0x0x0, motion/motion.c:535-552,641,753), slab-decomposed
    over the ranks of `group`.  `uniform=True` fuses motion's uniform-range scaling (motion.c:644-647 forward, :748-751 inverse)
    into the passes; the inverse additionally applies 1/(8 d h w).

    Layout.  Rank r owns frames block_range(d, r, G) of the clip ([dl, h, w], all rows) before the forward transform and rows
    block_range(h, r, G) of the coefficients afterwards ([d, hl, w], all frames).  d and h need NOT divide by G (BASELINE config 5's
    chroma planes are 960 x 540: 540 = 7 x 68 + 64 on 8 ranks): blocks are ceil-sized, the last one short.

    No pack / unpack sweeps.  The x pass is the LAST local pass of the forward transform and the FIRST after the inverse
    exchange, and it is separable by lines, so it writes (reads) the exchange buffers directly through its batch strides:
      forward   y pass in place on [dl, h, w]  ->  x pass of rows [r*hp + c0, ...) of every frame, straight into
                send[r] = [dlp, ch, w]  ->  all_to_all_single  ->  recv = [G*dlp, ch, w] = every frame, my rows  ->  z pass from
                recv (frame stride ch*w) into coeffs [d, hl, w]
      inverse   z pass from coeffs into send = [G*dlp, ch, w]  ->  all_to_all_single  ->  x pass from recv[s] = [dlp, ch, w]
                straight into rows [s*hp + c0, ...) of [dl, h, w]  ->  y pass in place
    Pipelined: the rows of a block are cut into `chunks` pieces; the exchange of piece p (RCCL, its own stream) runs while the x pass
    of piece p+1 and the z pass of piece p-1 run on the compute stream.  Two all-to-alls per roundtrip, no all-reduce; xGMI is
    point-to-point and an all-to-all uses all 7 links at once."""

    def __init__(self, d, h, w, group=None, uniform=True, lib=None, chunks=None):
        self.group = group
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        G = self.G
        self.d, self.h, self.w = d, h, w
        self.dlp, self.hp = -(-d // G), -(-h // G)               # block sizes (frames, rows)
        self.f_lo, self.f_hi = block_range(d, self.rank, G)
        self.y_lo, self.y_hi = block_range(h, self.rank, G)
        self.dl, self.hl = self.f_hi - self.f_lo, self.y_hi - self.y_lo
        if chunks is None:
            chunks = 1 if G == 1 else 4
        self.ch = -(-self.hp // max(1, min(chunks, self.hp)))    # rows per piece
        self.P = -(-self.hp // self.ch)
        self.lib = lib
        self.uniform = uniform
        self._plans = {}
        self._bufs = {}
        r2 = math.sqrt(2.0)
        u = (lambda p, fwd: p.set_axis_scale0(0, 1.0, 1.0 / r2) if fwd else p.set_axis_scale0(0, r2, 1.0)) if uniform else (lambda p, fwd: p)
        self._u = u
        dl, hl = self.dl, self.hl
        # y pass, in place on my frames: transform dim (h, stride w), batch (w, 1) x (dl, h*w)
        self.fwd_y = self.inv_y = None
        if dl:
            self.fwd_y = u(Plan.guru([(h, w, w)], [(w, 1, 1), (dl, h * w, h * w)], [REDFT10], lib=lib), True)
            self.inv_y = u(Plan.guru([(h, w, w)], [(w, 1, 1), (dl, h * w, h * w)], [REDFT01], lib=lib), False).set_scale(1.0 / (8.0 * d * h * w))

    # ---- plans that depend on a piece's row count (cached: at most three distinct counts occur) ----
    def _plan_x(self, nrows, fwd, nblocks=1):
        """x pass over `nrows` rows of every one of my frames, for `nblocks` consecutive destination (source) ranks in ONE launch: between
        the frame layout [dl, h, w] (the blocks' rows hp rows apart) and the blocks [dlp, ch, w] of an exchange buffer (dlp*ch*w apart).
        Round 3 launched one execution per rank: at 8 ranks x 4 pieces 32 launches of ~34 rows x 32 frames per plane and direction,
        launch-bound before xGMI is; the third batch level of the row passes (PassGeom::sb2_*) makes it one per piece."""
        key = ("x", nrows, fwd, nblocks)
        if key not in self._plans:
            w, h, ch = self.w, self.h, self.ch
            img, blk = (nblocks, self.hp * w), (nblocks, self.dlp * ch * w)
            if fwd:
                dims = [(nrows, w, w), (self.dl, h * w, ch * w)] + ([(img[0], img[1], blk[1])] if nblocks > 1 else [])
                p = Plan.guru([(w, 1, 1)], dims, [REDFT10], lib=self.lib)
            else:
                dims = [(nrows, w, w), (self.dl, ch * w, h * w)] + ([(img[0], blk[1], img[1])] if nblocks > 1 else [])
                p = Plan.guru([(w, 1, 1)], dims, [REDFT01], lib=self.lib)
            self._plans[key] = self._u(p, fwd)
        return self._plans[key]

    def _groups(self, p):
        """runs of consecutive blocks whose piece p has the same number of rows: [(first block, blocks, rows)] -- all full blocks form
        one run, a short last block its own, empty blocks none: at most two x-pass launches per piece whatever G is"""
        out = []
        for r in range(self.G):
            y0, y1 = self._rows_of(r, p)
            n = y1 - y0
            if n <= 0:
                continue
            if out and out[-1][2] == n and out[-1][0] + out[-1][1] == r:
                out[-1] = (out[-1][0], out[-1][1] + 1, n)
            else:
                out.append((r, 1, n))
        return out

    def _plan_z(self, nrows, fwd):
        """z pass over the columns of `nrows` of my rows, between a received piece [G*dlp, ch, w] and coeffs [d, hl, w]"""
        key = ("z", nrows, fwd)
        if key not in self._plans:
            w, ch, hl, d = self.w, self.ch, self.hl, self.d
            r2 = math.sqrt(2.0)
            if fwd:
                p = Plan.guru([(d, ch * w, hl * w)], [(nrows * w, 1, 1)], [REDFT10], lib=self.lib)
                p = self._u(p, True)
                if self.uniform:
                    p.set_scale(2.0 * r2)
            else:
                p = Plan.guru([(d, hl * w, ch * w)], [(nrows * w, 1, 1)], [REDFT01], lib=self.lib)
                p = self._u(p, False)
                if self.uniform:
                    p.set_scale(1.0 / (2.0 * r2))
            self._plans[key] = p
        return self._plans[key]

    def _buf(self, name, like):
        n = self.P * self.G * self.dlp * self.ch * self.w
        b = self._bufs.get(name)
        if b is None or b.device != like.device:
            b = torch.zeros(n, dtype=torch.float32, device=like.device)
            self._bufs[name] = b
        return b.view(self.P, self.G, self.dlp, self.ch, self.w)

    @staticmethod
    def _stream(t):
        return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0

    def _rows_of(self, r, p):
        """image rows of block r that fall into piece p: [y0, y1)"""
        lo, hi = block_range(self.h, r, self.G)
        y0 = min(hi, lo + p * self.ch)
        return y0, min(hi, y0 + self.ch)

    def _exchange(self, send, recv):
        if self.G == 1:
            return None                             # one rank: forward() / inverse() alias recv to send
        return dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group, async_op=True)

    def exchange_alone(self, like):
        """the P all-to-alls of one forward() (equally of one inverse()) with no pass beside them and each waited for: what the bench derives the
        exchange time and the xGMI fraction from.  `like`: any tensor on this rank's device.  Returns the bytes this rank sends to
        OTHER ranks in them (its own piece of every all-to-all stays local)."""
        send = self._buf("send", like)
        recv = send if self.G == 1 else self._buf("recv", like)
        for p in range(self.P):
            w = self._exchange(send[p], recv[p])
            if w is not None:
                w.wait()
        return send.numel() * 4 * (self.G - 1) // self.G

    def forward(self, frames):
        """frames: [dl, h, w] f32 (this rank's frames, block_range(d, rank, G); overwritten).  Returns [d, hl, w] coefficients of
        REDFT10^3 (uniform range if requested) for this rank's rows block_range(h, rank, G)."""
        assert tuple(frames.shape) == (self.dl, self.h, self.w) and frames.dtype == torch.float32 and frames.is_contiguous()
        st = self._stream(frames)
        send = self._buf("send", frames)
        recv = send if self.G == 1 else self._buf("recv", frames)
        coeffs = torch.empty((self.d, self.hl, self.w), dtype=torch.float32, device=frames.device)
        if self.dl:
            self.fwd_y.execute(frames.data_ptr(), stream=st)
        es = 4
        work = [None] * self.P

        def zpass(p):
            if work[p] is not None:
                work[p].wait()                      # RCCL: the compute stream waits for the collective; gloo: the host does
            y0, y1 = self._rows_of(self.rank, p)
            if y1 > y0:
                self._plan_z(y1 - y0, True).execute(recv[p].data_ptr(), coeffs.data_ptr() + (y0 - self.y_lo) * self.w * es, stream=st)

        for p in range(self.P):
            for r0, nb, n in (self._groups(p) if self.dl else []):
                y0 = self._rows_of(r0, p)[0]
                self._plan_x(n, True, nb).execute(frames.data_ptr() + y0 * self.w * es, send[p, r0].data_ptr(), stream=st)
            work[p] = self._exchange(send[p], recv[p])
            if p > 0:
                zpass(p - 1)
        zpass(self.P - 1)
        return coeffs

    def inverse(self, coeffs):
        """coeffs: [d, hl, w] (this rank's rows; not modified).  Returns this rank's [dl, h, w] frames, scaled by 1/(8dhw)."""
        assert tuple(coeffs.shape) == (self.d, self.hl, self.w) and coeffs.is_contiguous()
        st = self._stream(coeffs)
        send = self._buf("send", coeffs)
        recv = send if self.G == 1 else self._buf("recv", coeffs)
        frames = torch.empty((self.dl, self.h, self.w), dtype=torch.float32, device=coeffs.device)
        es = 4
        work = [None] * self.P

        def xpass(p):
            if work[p] is not None:
                work[p].wait()
            for s0, nb, n in (self._groups(p) if self.dl else []):
                y0 = self._rows_of(s0, p)[0]
                self._plan_x(n, False, nb).execute(recv[p, s0].data_ptr(), frames.data_ptr() + y0 * self.w * es, stream=st)

        for p in range(self.P):
            y0, y1 = self._rows_of(self.rank, p)
            if y1 > y0:
                self._plan_z(y1 - y0, False).execute(coeffs.data_ptr() + (y0 - self.y_lo) * self.w * es, send[p].data_ptr(), stream=st)
            work[p] = self._exchange(send[p], recv[p])
            if p > 0:
                xpass(p - 1)
        xpass(self.P - 1)
        if self.dl:
            self.inv_y.execute(frames.data_ptr(), stream=st)
        return frames


class ChannelShardedScan:
    """BASELINE config 4's multi-GPU layout: scan's progressive reconstruction (scan/scan.c:292-298,377-383,421-459)
    with the colour planes of the interleaved image spread over the ranks (plane z on rank z % G; a rank may own
    several planes or none).  Planes are independent, so there is NO collective in the frame loop; `gather()` uses
    one all_gather to rebuild the interleaved sum on every rank (e.g. to encode the output frame).

    Each rank holds its planes planar ([h][w] f32): forward REDFT10^2 with the 1/(4wh) normalisation fused, zigzag
    frame ids, then per output frame one fused masked-accumulate execution (or the pruned path for tiny steps).
    A rank that owns EVERY plane (one GPU) keeps the image interleaved as the tool has it (scan.c:292-293) and runs one
    execution per frame for the three channels instead of three; `layout="planar"` forces the planar form there too."""

    def __init__(self, image_hwc, step, group=None, lib=None, layout="auto"):
        from . import _lib
        self.lib = lib or _lib.load()
        self.group = group
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.h, self.w, self.c = image_hwc.shape
        self.step = int(step)
        self.nframes = (self.w * self.h + self.step - 1) // self.step          # scan.c:347-348 with limit = w*h
        self.mine = [z for z in range(self.c) if z % self.G == self.rank]
        if layout not in ("auto", "planar"):
            raise ValueError("layout is 'auto' or 'planar'")
        self.interleaved = layout == "auto" and len(self.mine) == self.c
        self.nch = self.c if self.interleaved else 1             # channels of one execution
        dev = image_hwc.device
        if self.interleaved:
            self.coeffs = [image_hwc.contiguous().clone()]
            self.fwd = Plan.image(self.h, self.w, self.c, REDFT10, lib=lib).set_scale(1.0 / (4.0 * self.w * self.h))
            self.inv = Plan.image(self.h, self.w, self.c, REDFT01, lib=lib)
        else:
            self.coeffs = [image_hwc[:, :, z].contiguous().clone() for z in self.mine]
            self.fwd = Plan.many_r2r([self.h, self.w], [REDFT10] * 2, lib=lib).set_scale(1.0 / (4.0 * self.w * self.h))
            self.inv = Plan.many_r2r([self.h, self.w], [REDFT01] * 2, lib=lib)
        self.ids = torch.zeros(self.w * self.h, dtype=torch.int32, device=dev)
        st = SlabDCT3D._stream(image_hwc) or None     # every helper on torch's current stream, like the transforms
        self._check(self.lib.dspfft_scan_zigzag_frame_ids(self.ids.data_ptr(), self.w, self.h, self.step, st))
        # the owner ids stay the same over the frames: let the fused step skip the column tiles a frame does not touch without reading their ids
        self.inv.scan_prepare(self.ids.data_ptr(), self.nch, stream=st or 0)
        self.work = torch.empty_like(self.coeffs[0]) if self.coeffs else None
        self.sums = []
        for cz in self.coeffs:
            self.fwd.execute(cz.data_ptr(), stream=SlabDCT3D._stream(cz))
            s = torch.empty_like(cz)
            self._check(self.lib.dspfft_broadcast_dc(s.data_ptr(), cz.data_ptr(), self.w * self.h, self.nch, st))   # scan.c:377-383
            self.sums.append(s)
        self.frame = 0

    def _check(self, rc):
        if rc:
            raise RuntimeError(self.lib.dspfft_last_error().decode())

    def warm(self):
        """one step with a frame id that no coefficient carries: every column tile is skipped and zeros are added to the running sums, which
        stay as they are -- loads the step's kernels before a timed loop"""
        for cz, s in zip(self.coeffs, self.sums):
            self.inv.execute_masked_accumulate(cz.data_ptr(), self.work.data_ptr(), s.data_ptr(), self.ids.data_ptr(), self.nframes, self.nch,
                                               stream=SlabDCT3D._stream(cz))

    def next_frame(self):
        """adds the coefficients of output frame `self.frame` into this rank's running sums"""
        if self.frame >= self.nframes:
            return False
        for cz, s in zip(self.coeffs, self.sums):
            self.inv.execute_masked_accumulate(cz.data_ptr(), self.work.data_ptr(), s.data_ptr(), self.ids.data_ptr(), self.frame, self.nch,
                                               stream=SlabDCT3D._stream(cz))
        self.frame += 1
        return True

    def gather(self):
        """interleaved (h, w, c) running sum on every rank"""
        dev = self.ids.device
        if self.interleaved:
            return self.sums[0].clone()
        out = torch.empty((self.h, self.w, self.c), dtype=torch.float32, device=dev)
        if self.G == 1:
            for z, s in zip(self.mine, self.sums):
                out[:, :, z] = s
            return out
        per = (self.c + self.G - 1) // self.G
        send = torch.zeros((per, self.h, self.w), dtype=torch.float32, device=dev)
        for i, s in enumerate(self.sums):
            send[i] = s
        recv = [torch.empty_like(send) for _ in range(self.G)]
        dist.all_gather(recv, send, group=self.group)
        for z in range(self.c):
            out[:, :, z] = recv[z % self.G][z // self.G]
        return out
