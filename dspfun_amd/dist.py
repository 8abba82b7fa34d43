"""Multi-GPU sharding of the hot path, one process per GPU over torch.distributed
(backend "nccl" == RCCL over xGMI on the MI355X node; "gloo" in the CPU tests).

Two regimes:

* independent units -- frames of motion's default `-b 0x0x1` (motion/motion.c:174,613-615), channel
  planes or whole images: `shard_range()` gives each rank a contiguous range; NO collective on the
  data path (bench.py uses this: weak scaling).

* one 3-D block spanning the whole clip (`-b 0x0x0`, motion/motion.c:535-552: REDFT10^3 / REDFT01^3 on a
  {d,h,w} block): `SlabDCT3D`.  Each rank owns d/G consecutive frames.  Forward = local 2-D passes
  (x, y) -> ONE all-to-all that re-slabs the volume by rows (each rank then owns h/G rows of ALL
  frames) -> local z pass.  The coefficients stay in the z-local layout, where motion's filters
  (elementwise, motion.c:650-744) can run; inverse = z pass -> all-to-all back -> 2-D passes.
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


class SlabDCT3D:
    """Distributed 3-D DCT-II / DCT-III of a {d,h,w} f32 volume, slab-decomposed over the ranks of
    `group`.  `uniform=True` fuses motion's uniform-range scaling (motion.c:644-647 forward,
    :748-751 inverse) into the passes; the inverse additionally applies 1/(8 d h w)."""

    def __init__(self, d, h, w, group=None, uniform=True, lib=None):
        self.group = group
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if d % self.G or h % self.G:
            raise ValueError(f"d={d} and h={h} must be divisible by the number of ranks ({self.G})")
        self.d, self.h, self.w = d, h, w
        self.dl, self.hl = d // self.G, h // self.G
        r2 = math.sqrt(2.0)
        # local 2-D passes over this rank's frames: rank 2 {h,w}, howmany = frames, dist = h*w
        self.fwd_yx = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=self.dl, idist=h * w, odist=h * w, lib=lib)
        self.inv_yx = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=self.dl, idist=h * w, odist=h * w, lib=lib)
        # z pass in the re-slabbed layout [d][hl*w]: rank 1 {d}, howmany = hl*w columns, stride = hl*w, dist = 1
        cols = self.hl * w
        self.fwd_z = Plan.many_r2r([d], [REDFT10], howmany=cols, istride=cols, idist=1, ostride=cols, odist=1, lib=lib)
        self.inv_z = Plan.many_r2r([d], [REDFT01], howmany=cols, istride=cols, idist=1, ostride=cols, odist=1, lib=lib)
        if uniform:
            for a in (0, 1):
                self.fwd_yx.set_axis_scale0(a, 1.0, 1.0 / r2)
                self.inv_yx.set_axis_scale0(a, r2, 1.0)
            self.fwd_z.set_axis_scale0(0, 1.0, 1.0 / r2).set_scale(2.0 * r2)
            self.inv_z.set_axis_scale0(0, r2, 1.0).set_scale(1.0 / (2.0 * r2))
        self.inv_yx.set_scale(1.0 / (8.0 * d * h * w))

    @staticmethod
    def _stream(t):
        return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0

    def _to_rows(self, x):
        """[dl, h, w] (my frames, all rows) -> [d, hl, w] (all frames, my rows): one all-to-all"""
        if self.G == 1:
            return x
        send = x.view(self.dl, self.G, self.hl, self.w).permute(1, 0, 2, 3).contiguous()
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        return recv.view(self.d, self.hl, self.w)

    def _to_frames(self, c):
        """[d, hl, w] -> [dl, h, w]: the inverse exchange"""
        if self.G == 1:
            return c
        send = c.view(self.G, self.dl, self.hl, self.w).contiguous()
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        return recv.permute(1, 0, 2, 3).contiguous().view(self.dl, self.h, self.w)

    def forward(self, frames):
        """frames: [d/G, h, w] f32 (this rank's frames; overwritten).  Returns [d, h/G, w] coefficients
        of REDFT10^3 (uniform range if requested) for this rank's row slab."""
        assert frames.shape == (self.dl, self.h, self.w) and frames.dtype == torch.float32 and frames.is_contiguous()
        self.fwd_yx.execute(frames.data_ptr(), stream=self._stream(frames))
        c = self._to_rows(frames)
        self.fwd_z.execute(c.data_ptr(), stream=self._stream(c))
        return c

    def inverse(self, coeffs):
        """coeffs: [d, h/G, w] (overwritten).  Returns this rank's [d/G, h, w] frames, scaled by 1/(8dhw)."""
        assert coeffs.shape == (self.d, self.hl, self.w) and coeffs.is_contiguous()
        self.inv_z.execute(coeffs.data_ptr(), stream=self._stream(coeffs))
        x = self._to_frames(coeffs)
        self.inv_yx.execute(x.data_ptr(), stream=self._stream(x))
        return x


class ChannelShardedScan:
    """BASELINE config 4's multi-GPU layout: scan's progressive reconstruction (scan/scan.c:292-298,377-383,421-459)
    with the colour planes of the interleaved image spread over the ranks (plane z on rank z % G; a rank may own
    several planes or none).  Planes are independent, so there is NO collective in the frame loop; `gather()` uses
    one all_gather to rebuild the interleaved sum on every rank (e.g. to encode the output frame).

    Each rank holds its planes planar ([h][w] f32): forward REDFT10^2 with the 1/(4wh) normalisation fused, zigzag
    frame ids, then per output frame one fused masked-accumulate execution (or the pruned path for tiny steps)."""

    def __init__(self, image_hwc, step, group=None, lib=None):
        from . import _lib
        self.lib = lib or _lib.load()
        self.group = group
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.h, self.w, self.c = image_hwc.shape
        self.step = int(step)
        self.nframes = (self.w * self.h + self.step - 1) // self.step          # scan.c:347-348 with limit = w*h
        self.mine = [z for z in range(self.c) if z % self.G == self.rank]
        dev = image_hwc.device
        self.coeffs = [image_hwc[:, :, z].contiguous().clone() for z in self.mine]
        self.fwd = Plan.many_r2r([self.h, self.w], [REDFT10] * 2, lib=lib).set_scale(1.0 / (4.0 * self.w * self.h))
        self.inv = Plan.many_r2r([self.h, self.w], [REDFT01] * 2, lib=lib)
        self.ids = torch.zeros(self.w * self.h, dtype=torch.int32, device=dev)
        self._check(self.lib.dspfft_scan_zigzag_frame_ids(self.ids.data_ptr(), self.w, self.h, self.step, None))
        self.work = torch.empty((self.h, self.w), dtype=torch.float32, device=dev)
        self.sums = []
        for cz in self.coeffs:
            self.fwd.execute(cz.data_ptr(), stream=SlabDCT3D._stream(cz))
            s = torch.empty_like(cz)
            self._check(self.lib.dspfft_broadcast_dc(s.data_ptr(), cz.data_ptr(), self.w * self.h, 1, None))   # scan.c:377-383
            self.sums.append(s)
        self.frame = 0

    def _check(self, rc):
        if rc:
            raise RuntimeError(self.lib.dspfft_last_error().decode())

    def next_frame(self):
        """adds the coefficients of output frame `self.frame` into this rank's running sums"""
        if self.frame >= self.nframes:
            return False
        for cz, s in zip(self.coeffs, self.sums):
            self.inv.execute_masked_accumulate(cz.data_ptr(), self.work.data_ptr(), s.data_ptr(), self.ids.data_ptr(), self.frame, 1,
                                               stream=SlabDCT3D._stream(cz))
        self.frame += 1
        return True

    def gather(self):
        """interleaved (h, w, c) running sum on every rank"""
        dev = self.ids.device
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
