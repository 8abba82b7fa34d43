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
