"""Host-side mirror of the FFTW plan/execute surface the reference uses (reference
include/precision.h:115 `fftw(call)`), over DEVICE memory.

    Plan.many_r2r(...)  <->  fftw(plan_many_r2r)   spec/spec.c:63  ispec.c:165  zoom.c:263  scan.c:292,359  motion.c:535-552
    Plan.r2r_2d(...)    <->  fftw(plan_r2r_2d)     applybasis/draw.c:74
    plan.execute(...)   <->  fftw(execute)
    plan.destroy()      <->  fftw(destroy_plan)

Pointers are raw device addresses (ints, e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t
handle as an int (torch.cuda.current_stream().cuda_stream) or 0 for the default stream.
"""
import ctypes as C

from . import _lib

REDFT01, REDFT10 = _lib.REDFT01, _lib.REDFT10


class DspfftError(RuntimeError):
    pass


def _ia(v):
    return None if v is None else (C.c_int * len(v))(*[int(x) for x in v])


def set_plan_effort(effort, lib=None):
    """dspfft_set_plan_effort: > 0 lets the plans made afterwards compile kernels for frame sizes spec_list.h does not list (FFTW_MEASURE's
    meaning: the plan will run many times); 0 (default) plans at once on the runtime-geometry kernels"""
    (lib or _lib.load()).dspfft_set_plan_effort(int(effort))


class Plan:
    """dtype "f32" (default; the fftwf_ API, COEFF_PRECISION=F) or "f64" (the fftw_ API of spec's, zoom's and
    applybasis's default build, include/precision.h:50-53): buffers, arithmetic and fused scales in that type."""

    def __init__(self, handle, lib, f64=False):
        self._h = handle
        self._lib = lib
        self.f64 = f64

    @classmethod
    def many_r2r(cls, n, kinds, howmany=1, inembed=None, istride=1, idist=0, onembed=None, ostride=1, odist=0, lib=None, dtype="f32",
                 first_axis_first=False):
        """first_axis_first: run axis 0 first and the contiguous axis last (f32 only); the inverse half of
        `roundtrip` is created this way"""
        if dtype not in ("f32", "f64"):
            raise ValueError("dtype must be 'f32' or 'f64'")
        if first_axis_first and dtype != "f32":
            raise ValueError("first_axis_first is an f32 plan option")
        lib = lib or _lib.load()
        n = list(n)
        kinds = list(kinds)
        if len(kinds) != len(n):
            raise ValueError("one kind per transformed dimension")
        h = C.c_void_p()
        if first_axis_first:
            rc = lib.dspfft_plan_many_r2r_ordered(C.byref(h), len(n), _ia(n), howmany, _ia(inembed), istride, idist, _ia(onembed), ostride, odist, _ia(kinds), 1)
        else:
            make = lib.dspfft_plan_many_r2r_f64 if dtype == "f64" else lib.dspfft_plan_many_r2r
            rc = make(C.byref(h), len(n), _ia(n), howmany, _ia(inembed), istride, idist, _ia(onembed), ostride, odist, _ia(kinds))
        if rc:
            raise DspfftError(lib.dspfft_last_error().decode())
        return cls(h, lib, dtype == "f64")

    @classmethod
    def guru(cls, dims, howmany_dims, kinds, lib=None, dtype="f32"):
        """The shape of fftw_plan_guru_r2r: dims and howmany_dims are lists of (n, in_stride, out_stride) in elements.  One plan for
        e.g. every 8x8x8 block of a volume (motion --blocksize 8x8x8)."""
        lib = lib or _lib.load()
        def arr(d):
            a = (_lib.IoDim * max(1, len(d)))()
            for i, (n, is_, os_) in enumerate(d):
                a[i].n, a[i].is_, a[i].os = int(n), int(is_), int(os_)
            return a
        h = C.c_void_p()
        rc = lib.dspfft_plan_guru_r2r(C.byref(h), len(dims), arr(dims), len(howmany_dims), arr(howmany_dims), _ia(list(kinds)), 1 if dtype == "f64" else 0)
        if rc:
            raise DspfftError(lib.dspfft_last_error().decode())
        return cls(h, lib, dtype == "f64")

    @classmethod
    def r2r_2d(cls, n0, n1, kind0, kind1, lib=None):
        lib = lib or _lib.load()
        h = C.c_void_p()
        if lib.dspfft_plan_r2r_2d(C.byref(h), n0, n1, kind0, kind1):
            raise DspfftError(lib.dspfft_last_error().decode())
        return cls(h, lib)

    @classmethod
    def image(cls, h, w, c, kind, lib=None, dtype="f32"):
        """The image tools' plan: rank 2 {h,w}, howmany=c, stride=c, dist=1 (interleaved HWC)."""
        return cls.many_r2r([h, w], [kind, kind], howmany=c, istride=c, idist=1, ostride=c, odist=1, lib=lib, dtype=dtype)

    def set_scale(self, scale):
        if self.f64:
            self._check(self._lib.dspfft_plan_set_scale_f64(self._h, scale))
        else:
            self._check(self._lib.dspfft_plan_set_scale(self._h, scale))
        return self

    def set_axis_scale0(self, axis, in_scale0=1.0, out_scale0=1.0):
        if self.f64:
            self._check(self._lib.dspfft_plan_set_axis_scale0_f64(self._h, axis, in_scale0, out_scale0))
        else:
            self._check(self._lib.dspfft_plan_set_axis_scale0(self._h, axis, in_scale0, out_scale0))
        return self

    def set_input_window(self, axis, lo, hi):
        """promise that input samples of `axis` outside [lo, hi) are zero; True when the plan then skips reading (and needing) them"""
        rc = self._lib.dspfft_plan_set_input_window(self._h, axis, lo, hi)
        if rc < 0:
            raise DspfftError(self._lib.dspfft_last_error().decode())
        return bool(rc)

    def set_input_modulation(self, axis, d_mul, reversed_from=0):
        """dspfft_plan_set_input_modulation (on top of an input window): True when honoured"""
        rc = self._lib.dspfft_plan_set_input_modulation(self._h, axis, d_mul or None, int(reversed_from))
        if rc < 0:
            raise DspfftError(self._lib.dspfft_last_error().decode())
        return rc == 1

    def set_output_alternate(self, axis, on=True):
        """output sample j of `axis` times (-1)^j, fused into that axis's pass; True when the plan honours it"""
        rc = self._lib.dspfft_plan_set_output_alternate(self._h, axis, int(on))
        if rc < 0:
            raise DspfftError(self._lib.dspfft_last_error().decode())
        return bool(rc)

    def execute(self, d_in, d_out=None, stream=0):
        d_out = d_in if d_out is None else d_out
        run = self._lib.dspfft_execute_f64 if self.f64 else self._lib.dspfft_execute
        self._check(run(self._h, C.c_void_p(d_in), C.c_void_p(d_out), C.c_void_p(stream)))

    @property
    def num_passes(self):
        return int(self._lib.dspfft_plan_num_passes(self._h))

    def execute_pass(self, index, d_in, d_out=None, stream=0):
        d_out = d_in if d_out is None else d_out
        self._check(self._lib.dspfft_execute_pass(self._h, index, C.c_void_p(d_in), C.c_void_p(d_out), C.c_void_p(stream)))

    def execute_masked_accumulate(self, d_in, d_work, d_acc, d_ids=0, frame_id=0, elems_per_id=1, stream=0):
        """scan/scan.c:429-459 fused: d_acc += plan(d_in where ids == frame_id)"""
        run = self._lib.dspfft_execute_masked_accumulate_f64 if self.f64 else self._lib.dspfft_execute_masked_accumulate
        self._check(run(
            self._h, C.c_void_p(d_in), C.c_void_p(d_work), C.c_void_p(d_acc), C.c_void_p(d_ids or None), frame_id, elems_per_id, C.c_void_p(stream)))

    def execute_sum2(self, other, d_in, d_in_other, d_out, stream=0):
        """d_out = self(d_in) + other(d_in_other) (dspfft_execute_sum2: one launch for two one-axis row REDFT01 plans on the same kernel)"""
        self._check(self._lib.dspfft_execute_sum2(self._h, other._h, d_in, d_in_other, d_out, stream or None))

    def scan_prepare(self, d_ids=0, elems_per_id=1, stream=0):
        """owner ids that stay the same over a scan's frames: record each column tile's id range so that the fused step skips a tile
        outside the frame without reading its ids (dspfft_plan_scan_prepare); d_ids = 0 forgets"""
        self._check(self._lib.dspfft_plan_scan_prepare(self._h, C.c_void_p(d_ids or None), elems_per_id, C.c_void_p(stream)))
        return self

    @staticmethod
    def _filter_params(filter):
        if filter is None:
            return None
        fp = _lib.MotionFilterParams()
        fp.active = (C.c_int * 3)(*filter["active"]); fp.minbuf_hw = (C.c_int * 2)(*filter["minbuf_hw"]); fp.block_depth = int(filter["block_depth"])
        fp.band_begin = (C.c_int * 3)(*filter["band_begin"]); fp.band_end = (C.c_int * 3)(*filter["band_end"])
        fp.damp = filter.get("damp", 1.0); fp.boost = filter.get("boost", 1.0)
        fp.threshold_lo = filter.get("threshold_lo", 0.0); fp.threshold_hi = filter.get("threshold_hi", 0.0)
        fp.preserve_dc = filter.get("preserve_dc", 0); fp.grey_add = filter.get("grey_add", 0.0); fp.quantizer = filter.get("quantizer", 0.0)
        return fp

    def roundtrip(self, inv, d_in, d_out=None, filter=None, d_coded=0, stream=0):
        """motion/motion.c:641-753: self (REDFT10) -> filter -> inv (REDFT01, created with first_axis_first=True), the middle axis
        fused into one launch when both plans have a specialised column kernel.  filter: dict with the fields of
        dspfft_motion_filter_params, or None."""
        d_out = d_in if d_out is None else d_out
        fp = self._filter_params(filter)
        self._check(self._lib.dspfft_execute_roundtrip(self._h, inv._h, C.c_void_p(d_in), C.c_void_p(d_out), C.byref(fp) if fp is not None else None,
                                                       C.c_void_p(d_coded or None), C.c_void_p(stream)))

    def roundtrip_u8(self, inv, d_in_u8, d_out_u8, d_work, out_mul, filter=None, d_coded=0, stream=0):
        """the same with motion's 8-bit samples at both ends (motion.c:617-640, :760-776): u8 in, float work buffer, u8 out =
        quantise(value * out_mul); the conversions ride on the first and last row passes when those are planar specialised passes"""
        fp = self._filter_params(filter)
        self._check(self._lib.dspfft_execute_roundtrip_u8(self._h, inv._h, C.c_void_p(d_in_u8), C.c_void_p(d_out_u8), C.c_void_p(d_work), out_mul,
                                                          C.byref(fp) if fp is not None else None, C.c_void_p(d_coded or None), C.c_void_p(stream)))

    def describe(self):
        buf = C.create_string_buffer(4096)
        self._check(self._lib.dspfft_plan_describe(self._h, buf, len(buf)))
        return buf.value.decode()

    @property
    def algorithmic_bytes(self):
        return int(self._lib.dspfft_plan_algorithmic_bytes(self._h))

    def destroy(self):
        if self._h:
            self._lib.dspfft_destroy_plan(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise DspfftError(self._lib.dspfft_last_error().decode())


class Stream:
    """A HIP stream of the library's own (dspfft_stream_create: hipStreamNonBlocking).  `handle` is what execute / Batch take."""

    def __init__(self, lib=None):
        self._lib = lib or _lib.load()
        self.handle = self._lib.dspfft_stream_create()
        if not self.handle:
            raise DspfftError("stream creation failed")

    def synchronize(self):
        if self._lib.dspfft_stream_synchronize(self.handle):
            raise DspfftError(self._lib.dspfft_last_error().decode())

    def __del__(self):
        try:
            self._lib.dspfft_stream_destroy(self.handle)
        except Exception:
            pass


class Events:
    """n timing events of the library (dspfft_event_create): recorded by Batch.run on the streams of the bracketed items"""

    def __init__(self, n, lib=None):
        self._lib = lib or _lib.load()
        self.handles = (C.c_void_p * n)(*[self._lib.dspfft_event_create() for _ in range(n)])
        if any(h is None for h in self.handles):
            raise DspfftError("event creation failed")

    def elapsed_ms(self, i, j):
        ms = C.c_float()
        if self._lib.dspfft_event_elapsed_ms(self.handles[i], self.handles[j], C.byref(ms)):
            raise DspfftError(self._lib.dspfft_last_error().decode())
        return float(ms.value)

    def __del__(self):
        try:
            for h in self.handles:
                self._lib.dspfft_event_destroy(h)
        except Exception:
            pass


class Batch:
    """A fixed list of executions -- (plan, d_in, d_out, stream) per item -- enqueued by ONE call of dspfft_execute_many: the
    per-frame loop of motion (motion/motion.c:613-753) or of a clip of spec/ispec frames, without a binding-layer call per frame."""

    def __init__(self, items, lib=None):
        items = list(items)
        self._lib = lib or _lib.load()
        self._keep = [it[0] for it in items]
        n = len(items)
        self.n = n
        self._plans = (C.c_void_p * n)(*[it[0]._h for it in items])
        self._in = (C.c_void_p * n)(*[it[1] for it in items])
        self._out = (C.c_void_p * n)(*[(it[1] if it[2] is None else it[2]) for it in items])
        self._streams = (C.c_void_p * n)(*[(it[3] or None) for it in items])

    def run(self, timed_item=0, timed_count=0, events=None, event_offset=0):
        ev = None
        if events is not None and timed_count:
            ev = C.cast(C.byref(events.handles, event_offset * C.sizeof(C.c_void_p)), C.POINTER(C.c_void_p))
        rc = self._lib.dspfft_execute_many(self.n, self._plans, self._in, self._out, self._streams, timed_item, timed_count if ev is not None else 0, ev)
        if rc:
            raise DspfftError(self._lib.dspfft_last_error().decode())

    def run_repeat(self, repeats, rejoin_every=0, timed_every=0, timed_count=0, events=None):
        """dspfft_execute_many_repeat: the batch `repeats` times in one library call (the frame loop of a clip); the streams of the
        batch are re-joined every `rejoin_every` repeats; every `timed_every`-th repeat brackets the passes of a rotating window of
        `timed_count` items with `events` (2 per pass, in order)."""
        ev = None
        if events is not None and timed_every and timed_count:
            ev = C.cast(events.handles, C.POINTER(C.c_void_p))
        rc = self._lib.dspfft_execute_many_repeat(self.n, self._plans, self._in, self._out, self._streams, int(repeats), int(rejoin_every),
                                                  int(timed_every) if ev is not None else 0, int(timed_count) if ev is not None else 0, ev)
        if rc:
            raise DspfftError(self._lib.dspfft_last_error().decode())

