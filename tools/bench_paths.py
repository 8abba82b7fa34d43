"""Secondary measurements (not the headline): zoom C3 (MFMA), scan C4 fused frame step, motion C5 3-D.
Prints one JSON object.  Run on the GPU box: python tools/bench_paths.py"""
import ctypes as C, json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import Plan, _lib, REDFT10, REDFT01
from dspfun_amd.zoom import Zoom

L = _lib.load()
dev = "cuda:0"
res = {}


def timeit(fn, reps=10, warm=2):
    """events around `reps` calls -- at least 40 ms of them: ten calls of a 0.3 ms roundtrip end before the clocks have settled (the double
    4K roundtrip read 0.374 ms that way and 0.32-0.33 over 200 calls)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    est = a.elapsed_time(b) / reps
    more = min(400, int(40.0 / max(est, 1e-3)))
    if more <= reps:
        return est
    a.record()
    for _ in range(more):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / more


# ---- zoom, config 3: 1920x1080 -> 7680x4320 ----
w, h = 1920, 1080
z = Zoom(torch, torch.rand(h, w, 3, device=dev))
ms = timeit(lambda: z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), method="gemm"), reps=5, warm=1)
flop = 2.0 * 3 * (h * (4 * w) * w + (4 * h) * (4 * w) * h)
res["zoom_c3"] = {"ms_per_frame": round(ms, 3), "TFLOPs": round(flop / ms / 1e9, 2), "GFLOP": round(flop / 1e9, 1),
                  "mfma_f32_peak_TFLOPs": 157.3, "frac": round(flop / ms / 1e9 / 157.3, 4), "note": "dense MFMA product (method gemm); includes basis generation + deinterleave"}
ms = timeit(lambda: z.frame(4 * w, 4 * h, (4.0, 1.0), (4.0, 1.0), method="fft"), reps=10, warm=2)
# compulsory bytes of the frame: the coefficients in, the 7680x4320x3 frame out
res["zoom_c3_fft"] = {"ms_per_frame": round(ms, 3), "speedup_over_gemm": round(res["zoom_c3"]["ms_per_frame"] / ms, 2),
                      "algorithmic_GBps": round((w * h * 3 + 16 * w * h * 3) * 4 / ms / 1e6, 1), "frac_of_8TBps": round((w * h * 3 + 16 * w * h * 3) * 4 / ms / 1e6 / 8000, 4),
                      "note": "dspfft_zoomfft_*: y first on the coefficients' columns (two windowed column passes), x last on the duo row kernel (dspfft_cosrows_*): cosine and sine part as a packed pair through one half-length FFT per channel"}
# the same source at 3.7x with the centered basis (off the DCT-III grid): chirp-z transforms against the dense product (round 4)
vw37, vh37 = int(w * 3.7), int(h * 3.7)
msg = timeit(lambda: z.frame(vw37, vh37, (3.7, 1.0), (3.7, 1.0), 12.5, -4.25, 1, method="gemm"), reps=5, warm=1)
msc = timeit(lambda: z.frame(vw37, vh37, (3.7, 1.0), (3.7, 1.0), 12.5, -4.25, 1, method="czt"), reps=10, warm=2)
res["zoom_1080p_3p7x_centered"] = {"czt_ms_per_frame": round(msc, 3), "gemm_ms_per_frame": round(msg, 3), "speedup": round(msg / msc, 2),
                                   "note": "dspfft_zoomczt_*: Bluestein convolutions of 5400 (y) and 9600 (x) points in LDS, two transposes between"}
del z

# ---- 8K double frame roundtrip (spec / zoom's default COEFF_PRECISION=D build at configs 3 / 4's frame size) ----
w, h, c = 7680, 4320, 3
x8d = torch.rand(h, w, c, device=dev, dtype=torch.float64)
f8d = Plan.image(h, w, c, REDFT10, dtype="f64")
i8d = Plan.image(h, w, c, REDFT01, dtype="f64").set_scale(1.0 / (4.0 * w * h))
def rt8d():
    f8d.execute(x8d.data_ptr()); i8d.execute(x8d.data_ptr())
ms = timeit(rt8d, reps=5, warm=2)
res["f64_8k_frame_roundtrip"] = {"ms": round(ms, 3), "Mpixels_per_s": round(h * w / ms / 1e3, 1), "frac_of_8TBps": round(h * w * c * 32 / ms / 1e6 / 8000, 4), "plan": f8d.describe()}
del x8d, f8d, i8d

# ---- 8K frame roundtrip (the transform of configs 3 and 4's frame size) ----
w, h, c = 7680, 4320, 3
x8 = torch.rand(h, w, c, device=dev)
f8 = Plan.image(h, w, c, REDFT10)
i8 = Plan.image(h, w, c, REDFT01).set_scale(1.0 / (4.0 * w * h))
def rt8():
    f8.execute(x8.data_ptr()); i8.execute(x8.data_ptr())
ms = timeit(rt8, reps=10, warm=2)
res["f32_8k_frame_roundtrip"] = {"ms": round(ms, 3), "Mpixels_per_s": round(h * w / ms / 1e3, 1), "frac_of_8TBps": round(h * w * c * 16 / ms / 1e6 / 8000, 4), "plan": f8.describe()}
del x8, f8, i8
# ---- scan, config 4: 7680x4320x3, zigzag, step 2^20 -> 32 frames, fused step ----
w, h, c = 7680, 4320, 3
coeffs = torch.rand(h, w, c, device=dev)
Plan.image(h, w, c, REDFT10).set_scale(1.0 / (4.0 * w * h)).execute(coeffs.data_ptr())
inv = Plan.image(h, w, c, REDFT01)
ids = torch.zeros(w * h, dtype=torch.int32, device=dev)
step = 1 << 20
nframes = (w * h + step - 1) // step
L.dspfft_scan_zigzag_frame_ids(ids.data_ptr(), w, h, step, None)
inv.scan_prepare(ids.data_ptr(), c)          # as host/scan_dev.c does for owner ids that stay the same over the frames
acc = torch.empty_like(coeffs); work = torch.empty_like(coeffs); recon = torch.empty_like(coeffs); image = torch.empty_like(coeffs)
order = torch.zeros(w * h, dtype=torch.int32, device=dev)
L.dspfft_scan_zigzag(order.data_ptr(), w, h, 0, w * h, None)
f = [0]
def fused():
    inv.execute_masked_accumulate(coeffs.data_ptr(), work.data_ptr(), acc.data_ptr(), ids.data_ptr(), f[0] % nframes, c); f[0] += 1
def unfused():
    k = f[0] % nframes; f[0] += 1
    first = k * step; cnt = min(step, w * h - first)
    L.dspfft_scan_scatter(recon.data_ptr(), coeffs.data_ptr(), order.data_ptr() + 4 * first, cnt, w * h, c, None)
    inv.execute(recon.data_ptr(), image.data_ptr())
    L.dspfft_accumulate(acc.data_ptr(), image.data_ptr(), w * h * c, None)
msf = timeit(fused, reps=8); msu = timeit(unfused, reps=8)
samples = w * h * c
res["scan_c4_frame_step"] = {"fused_ms": round(msf, 3), "unfused_ms": round(msu, 3), "algorithmic_GBps_fused": round(samples * 12 / msf / 1e6, 1),
                             "frac_of_8TBps": round(samples * 12 / msf / 1e6 / 8000, 4), "frames": nframes, "note": "12 B/sample algorithmic (SURVEY 8d); 1 GPU, all 3 channels"}

del coeffs, acc, work, recon, image, ids, order
# ---- motion, config 5 luma plane: 1920x1080x256 3-D roundtrip ----
d_, h, w = 256, 1080, 1920
vol = torch.rand(d_, h, w, device=dev)
r2 = math.sqrt(2.0)
fwd = Plan.many_r2r([d_, h, w], [REDFT10] * 3).set_scale(2 * r2)
invp = Plan.many_r2r([d_, h, w], [REDFT01] * 3).set_scale(1.0 / (2 * r2) / (8.0 * d_ * h * w))
for a in range(3):
    fwd.set_axis_scale0(a, 1.0, 1.0 / r2); invp.set_axis_scale0(a, r2, 1.0)
def rt():
    fwd.execute(vol.data_ptr()); invp.execute(vol.data_ptr())
ms = timeit(rt, reps=3, warm=1)
n = d_ * h * w
res["motion_c5_luma_3d_roundtrip"] = {"ms": round(ms, 2), "Msamples_per_s": round(n / ms / 1e3, 1), "algorithmic_GBps": round(n * 16 / ms / 1e6, 1),
                                      "frac_of_8TBps": round(n * 16 / ms / 1e6 / 8000, 4), "plan": fwd.describe()}
# per-frame 2-D (default -b 0x0x1) over the same 256 frames, one batched plan
f2 = Plan.many_r2r([h, w], [REDFT10] * 2, howmany=d_, idist=h * w, odist=h * w)
i2 = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=d_, idist=h * w, odist=h * w).set_scale(1.0 / (4.0 * h * w))
def rt2():
    f2.execute(vol.data_ptr()); i2.execute(vol.data_ptr())
ms = timeit(rt2, reps=3, warm=1)
res["motion_c5_luma_per_frame_2d_roundtrip"] = {"ms": round(ms, 2), "Msamples_per_s": round(n / ms / 1e3, 1), "algorithmic_GBps": round(n * 16 / ms / 1e6, 1),
                                                "frac_of_8TBps": round(n * 16 / ms / 1e6 / 8000, 4)}
# the same through dspfft_execute_roundtrip (inverse in first-axis-first order, middle axis fused) with motion's quantiser
import os as _os
flt = dict(active=(d_, h, w), minbuf_hw=(h, w), block_depth=d_, band_begin=(0, 0, 0), band_end=(d_, h, w), quantizer=3.0)
inv3 = Plan.many_r2r([d_, h, w], [REDFT01] * 3, first_axis_first=True).set_scale(1.0 / (2 * r2) / (8.0 * d_ * h * w))
for a in range(3):
    inv3.set_axis_scale0(a, r2, 1.0)
coded = torch.zeros(1, dtype=torch.int64, device=dev)
I3, I2 = (C.c_int * 3), (C.c_int * 2)
def rt_unfused_filter():
    fwd.execute(vol.data_ptr())
    L.dspfft_motion_filter(vol.data_ptr(), I3(d_, h, w), I2(h, w), I3(0, 0, 0), I3(d_, h, w), 1.0, 1.0, 0.0, 0.0, 0, 0.0, 3.0, coded.data_ptr(), None)
    invp.execute(vol.data_ptr())
vol.copy_(torch.rand(d_, h, w, device=dev) * 255)
ms_u = timeit(rt_unfused_filter, reps=3, warm=1)
vol.copy_(torch.rand(d_, h, w, device=dev) * 255)
ms_f = timeit(lambda: fwd.roundtrip(inv3, vol.data_ptr(), filter=flt, d_coded=coded.data_ptr()), reps=3, warm=1)
res["motion_c5_luma_3d_filtered_roundtrip"] = {"unfused_ms": round(ms_u, 2), "fused_ms": round(ms_f, 2), "speedup": round(ms_u / ms_f, 2),
                                               "fused_algorithmic_GBps": round(n * 40 / ms_f / 1e6, 1), "note": "fused: 5 launches, 40 B/sample; unfused: 7 launches, 56 B/sample"}
flt2 = dict(active=(1, h, w), minbuf_hw=(h, w), block_depth=1, band_begin=(0, 0, 0), band_end=(1, h, w), quantizer=3.0)
i2r = Plan.many_r2r([h, w], [REDFT01] * 2, howmany=d_, idist=h * w, odist=h * w, first_axis_first=True).set_scale(1.0 / (4.0 * h * w))
vol.copy_(torch.rand(d_, h, w, device=dev) * 255)
ms_f2 = timeit(lambda: f2.roundtrip(i2r, vol.data_ptr(), filter=flt2, d_coded=coded.data_ptr()), reps=3, warm=1)
vol.copy_(torch.rand(d_, h, w, device=dev) * 255)
_os.environ["DSPFFT_NO_FUSED_ROUNDTRIP"] = "1"
ms_u2 = timeit(lambda: f2.roundtrip(i2r, vol.data_ptr(), filter=flt2, d_coded=coded.data_ptr()), reps=3, warm=1)
del _os.environ["DSPFFT_NO_FUSED_ROUNDTRIP"]
res["motion_c5_luma_per_frame_filtered_roundtrip"] = {"unfused_ms": round(ms_u2, 2), "fused_ms": round(ms_f2, 2), "speedup": round(ms_u2 / ms_f2, 2),
                                                      "fused_algorithmic_GBps": round(n * 24 / ms_f2 / 1e6, 1), "note": "256 frames of 1920x1080, fused: 3 launches, 24 B/sample; unfused: 5 launches, 40 B/sample"}
# ---- motion end to end on 8-bit frames (motion.c:617-776): u8 load, transform, quantise, inverse, u8 store ----
v8 = (torch.rand(d_, h, w, device=dev) * 255).to(torch.uint8)
o8 = torch.empty_like(v8)
def motion_unfused_3d():
    L.dspfft_u8_to_f32(vol.data_ptr(), v8.data_ptr(), n, None)
    fwd.execute(vol.data_ptr())
    L.dspfft_motion_filter(vol.data_ptr(), I3(d_, h, w), I2(h, w), I3(0, 0, 0), I3(d_, h, w), 1.0, 1.0, 0.0, 0.0, 0, 0.0, 3.0, coded.data_ptr(), None)
    invp.execute(vol.data_ptr())
    L.dspfft_f32_to_u8(o8.data_ptr(), vol.data_ptr(), 1.0, n, None)
ms_u = timeit(motion_unfused_3d, reps=3, warm=1)
ms_f = timeit(lambda: fwd.roundtrip_u8(inv3, v8.data_ptr(), o8.data_ptr(), vol.data_ptr(), 1.0, filter=flt, d_coded=coded.data_ptr()), reps=3, warm=1)
res["motion_c5_luma_3d_u8_to_u8"] = {"separate_steps_ms": round(ms_u, 2), "fused_ms": round(ms_f, 2), "speedup": round(ms_u / ms_f, 2),
                                     "Msamples_per_s": round(n / ms_f / 1e3, 1), "note": "separate: 9 launches, 66 B/sample; fused: 5 launches, 34 B/sample"}
def motion_unfused_2d():
    L.dspfft_u8_to_f32(vol.data_ptr(), v8.data_ptr(), n, None)
    f2.execute(vol.data_ptr())
    L.dspfft_motion_filter(vol.data_ptr(), I3(d_, h, w), I2(h, w), I3(0, 0, 0), I3(d_, h, w), 1.0, 1.0, 0.0, 0.0, 0, 0.0, 3.0, coded.data_ptr(), None)
    i2.execute(vol.data_ptr())
    L.dspfft_f32_to_u8(o8.data_ptr(), vol.data_ptr(), 1.0, n, None)
ms_u2 = timeit(motion_unfused_2d, reps=3, warm=1)
ms_f2 = timeit(lambda: f2.roundtrip_u8(i2r, v8.data_ptr(), o8.data_ptr(), vol.data_ptr(), 1.0, filter=flt2, d_coded=coded.data_ptr()), reps=3, warm=1)
res["motion_c5_luma_per_frame_u8_to_u8"] = {"separate_steps_ms": round(ms_u2, 2), "fused_ms": round(ms_f2, 2), "speedup": round(ms_u2 / ms_f2, 2),
                                            "Msamples_per_s": round(n / ms_f2 / 1e3, 1), "fps_1080p_luma": round(d_ / ms_f2 * 1e3, 0),
                                            "note": "256 frames; separate: 7 launches, 50 B/sample; fused: 3 launches, 18 B/sample"}
del vol, v8, o8
# ---- the same luma volume through SlabDCT3D on this one GPU (the multi-GPU path's local work: y, x passes into the exchange buffers, z pass) ----
from dspfun_amd.dist import SlabDCT3D
eng = SlabDCT3D(d_, h, w, chunks=4)
v3 = (torch.rand(d_, h, w, device=dev) * 255).floor()
ms = timeit(lambda: eng.inverse(eng.forward(v3)), reps=3, warm=1)
res["motion_c5_luma_slab3d_1gpu"] = {"ms": round(ms, 2), "algorithmic_GBps": round(n * 16 / ms / 1e6, 1), "frac_of_8TBps": round(n * 16 / ms / 1e6 / 8000, 4), "row_pieces": eng.P}
del eng, v3
torch.cuda.empty_cache()
# ---- double precision (fftw_ API, spec's default build): 4K frame roundtrip on the runtime-geometry kernels ----
h, w, c = 2160, 3840, 3
x64 = torch.rand(h, w, c, device=dev, dtype=torch.float64)
f64f = Plan.image(h, w, c, REDFT10, dtype="f64").set_scale(1.0 / (4.0 * w * h))
f64i = Plan.image(h, w, c, REDFT01, dtype="f64")
def rt64():
    f64f.execute(x64.data_ptr()); f64i.execute(x64.data_ptr())
ms = timeit(rt64, reps=10, warm=2)
res["f64_c2_frame_roundtrip"] = {"ms": round(ms, 3), "Mpixels_per_s": round(h * w / ms / 1e3, 1), "algorithmic_GBps": round(h * w * c * 32 / ms / 1e6, 1),
                                 "frac_of_8TBps": round(h * w * c * 32 / ms / 1e6 / 8000, 4), "plan": f64f.describe()}
# the same frame in f32 with the specialised kernels disabled (DSPFFT_NO_SPEC=1 at plan time): the generic kernels' own speed
os.environ["DSPFFT_NO_SPEC"] = "1"
x32 = torch.rand(h, w, c, device=dev)
g32f = Plan.image(h, w, c, REDFT10).set_scale(1.0 / (4.0 * w * h))
g32i = Plan.image(h, w, c, REDFT01)
del os.environ["DSPFFT_NO_SPEC"]
def rt32g():
    g32f.execute(x32.data_ptr()); g32i.execute(x32.data_ptr())
ms = timeit(rt32g, reps=10, warm=2)
res["f32_generic_c2_frame_roundtrip"] = {"ms": round(ms, 3), "Mpixels_per_s": round(h * w / ms / 1e3, 1)}
print(json.dumps(res, indent=1))
