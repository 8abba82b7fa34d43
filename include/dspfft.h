/*
 * dspfft.h -- device-resident C ABI of the MI355X real-even DCT engine.
 *
 * This is the second-level boundary of SURVEY.md section 8(b): the same plan/execute contract the
 * reference reaches through `fftw(call)` (include/precision.h:115 of the reference), but on DEVICE
 * pointers and an explicit HIP stream, so buffers stay resident in HBM between transforms.
 * include/fftw3.h (the FFTW-named host-pointer shim the tools link against) is a thin adapter over
 * these entry points.
 *
 * All functions return 0 on success and a negative code on failure; dspfft_last_error() describes
 * the most recent failure of the calling thread.  No function touches user arrays at plan time
 * (scan/scan.c:354-359 relies on that, see SURVEY.md section 7 "FFTW_MEASURE clobbering").
 */
#ifndef DSPFFT_H
#define DSPFFT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* same numeric values as FFTW's fftw_r2r_kind for the two kinds the reference uses */
enum { DSPFFT_REDFT01 = 4, DSPFFT_REDFT10 = 5 };

typedef struct dspfft_plan_s *dspfft_plan;

/* Replaces fftw(plan_many_r2r) -- spec/spec.c:63, spec/ispec.c:165, zoom/zoom.c:263, scan/scan.c:292,359,
 * motion/motion.c:535-538,549-552.  rank 1..3, kinds[i] in {DSPFFT_REDFT10, DSPFFT_REDFT01}, f32 data.
 * Addressing follows FFTW's advanced interface: element (i0..i_{r-1}) of transform t is at
 * t*dist + stride*(((i0*embed[1] + i1)*embed[2] + i2)); embed == NULL means embed = n. */
int dspfft_plan_many_r2r(dspfft_plan *plan, int rank, const int *n, int howmany,
                         const int *inembed, int istride, int idist,
                         const int *onembed, int ostride, int odist, const int *kinds);

/* Replaces fftw(plan_r2r_2d) -- applybasis/draw.c:74. */
int dspfft_plan_r2r_2d(dspfft_plan *plan, int n0, int n1, int kind0, int kind1);

/* Fused per-index normalisation: every output is multiplied by `scale`; along
 * transformed axis `axis` (0-based, as in n[]) input index 0 is multiplied by in_scale0 before the
 * transform and output index 0 by out_scale0 after it.  Defaults: 1, 1, 1.
 *   spec/spec.c:70-78     = scale 1/(2wh), out_scale0 1/sqrt2 on both axes
 *   spec/ispec.c:153-159  = scale 1/2,     in_scale0  sqrt2   on both axes
 *   scan/scan.c:296-298   = scale 1/(4wh)
 *   motion/motion.c:644-647 (block == scaled) = scale 2 sqrt2, out_scale0 1/sqrt2 on all three axes */
int dspfft_plan_set_scale(dspfft_plan plan, float scale);
int dspfft_plan_set_axis_scale0(dspfft_plan plan, int axis, float in_scale0, float out_scale0);

/* Replaces fftw(execute) -- 9 call sites.  d_in/d_out are DEVICE pointers laid out as
 * the plan describes; d_in == d_out is the in-place case.  Asynchronous on `hip_stream`
 * (a hipStream_t; NULL = the default stream). */
int dspfft_execute(dspfft_plan plan, const float *d_in, float *d_out, void *hip_stream);

/* Profiling aid: a plan is a sequence of 1-D axis passes (one kernel launch each, last axis
 * first).  dspfft_execute_pass runs pass `index` alone, exactly as dspfft_execute would (pass 0
 * reads d_in, later passes work in place on d_out), so each kernel can be bracketed by events. */
int dspfft_plan_num_passes(dspfft_plan plan);
int dspfft_execute_pass(dspfft_plan plan, int index, const float *d_in, float *d_out, void *hip_stream);

/* Many executions with one call: item i runs plans[i] on (d_in[i], d_out[i]) on hip_streams[i] (f32 plans), in order.  This is what
 * a tool that loops fftw(execute) over the frames of a clip (motion/motion.c:613-753, scan/scan.c:447) calls once per batch, so a
 * binding layer (cgo, ctypes ...) pays its per-call cost once per batch instead of once per frame and pass.
 * Profiling aid: with pass_events non-NULL, every pass of items [timed_item, timed_item + timed_count) is bracketed by events
 * recorded on that item's stream: the j-th bracketed pass between pass_events[2j] and pass_events[2j+1] (handles from
 * dspfft_event_create; read with dspfft_event_elapsed_ms after dspfft_event_synchronize or a stream/device synchronise). */
int dspfft_execute_many(int count, const dspfft_plan *plans, const float *const *d_in, float *const *d_out, void *const *hip_streams,
                        int timed_item, int timed_count, void *const *pass_events);
/* The same batch `repeats` times with ONE call: the frame loop of a clip inside the library (motion/motion.c:613-753 executes its
 * per-frame plans once per frame, scan/scan.c:421-447 its inverse plan once per output frame), so nothing of the host language sits
 * between two frames.  rejoin_every > 0 and more than one distinct stream in the batch: every rejoin_every repeats each stream waits
 * for the point the others have reached (two free-running streams otherwise keep whatever relative phase the first repeat's host timing
 * gave them, and that phase decides how well their kernels share the CUs).
 * Profiling aid: with pass_events non-NULL and timed_every > 0, repeats 0, timed_every, 2 timed_every ... bracket every pass of a window
 * of timed_count consecutive items (timed_count must divide count); the window advances by timed_count items (mod count) each time, and
 * the events of the j-th bracketed pass overall are pass_events[2j], pass_events[2j+1].  One-pass block plans cannot be bracketed.
 * Items may be f32 or f64 plans: each item's buffers are float or double as its plan says (hence the void pointers). */
int dspfft_execute_many_repeat(int count, const dspfft_plan *plans, const void *const *d_in, void *const *d_out, void *const *hip_streams,
                               int repeats, int rejoin_every, int timed_every, int timed_count, void *const *pass_events);
/* Streams of the library's own for the calls above: plain non-blocking HIP streams (hipStreamCreateWithFlags(hipStreamNonBlocking)),
 * for hosts without a HIP binding of their own (cgo, ctypes ...) and for hosts whose framework hands out streams from a pool:
 * two streams of torch's pool were seen sharing a hardware queue, which serialises the two frames they carry (48K instead of 56K
 * Mpix/s on the 4K roundtrip, tools/py_enqueue_probe.py).  NULL on failure. */
void *dspfft_stream_create(void);
void dspfft_stream_destroy(void *stream);
int dspfft_stream_synchronize(void *stream);
void *dspfft_event_create(void);
void dspfft_event_destroy(void *event);
int dspfft_event_synchronize(void *event);
int dspfft_event_elapsed_ms(void *start, void *stop, float *ms);

/* Double-precision samples: the fftw_ (no suffix) API that spec, zoom and applybasis use in their default
 * build (COEFF_PRECISION ?= D: spec/Makefile:1, applybasis/Makefile:1; include/precision.h:50-53,66-72).
 * Same geometry and call-site contract as above with `double` buffers; arithmetic, twiddle tables and the
 * fused scales are double.  These plans run the runtime-geometry kernels (the compile-time-specialised
 * ones are f32 only).  The _f64 scale setters also work on f32 plans (rounded to float there);
 * dspfft_execute on an f64 plan, or dspfft_execute_f64 on an f32 plan, fails with an error. */
int dspfft_plan_many_r2r_f64(dspfft_plan *plan, int rank, const int *n, int howmany,
                             const int *inembed, int istride, int idist,
                             const int *onembed, int ostride, int odist, const int *kinds);
int dspfft_plan_set_scale_f64(dspfft_plan plan, double scale);
int dspfft_plan_set_axis_scale0_f64(dspfft_plan plan, int axis, double in_scale0, double out_scale0);
int dspfft_execute_f64(dspfft_plan plan, const double *d_in, double *d_out, void *hip_stream);
/* Optional: the caller promises that input samples of `axis` outside [lo, hi) are ZERO (a spectrum zero-padded to a longer transform:
 * zoom's y stage, motion's scaled > block) so that they need not be read.  Returns 1 when the plan honours it -- then those rows need
 * not even be stored -- and 0 when it does not (nothing changes: the zeros must really be there).  Today: f32 plans whose first pass
 * is a listed specialised column or row REDFT01 pass along `axis`.  lo = hi = 0 turns it off.  Negative: bad arguments. */
int dspfft_plan_set_input_window(dspfft_plan plan, int axis, int lo, int hi);
/* On top of an input window (set it first; setting a window again drops this): input sample x of `axis` is taken from position
 * p = reversed_from > 0 ? reversed_from - x : x of its line and multiplied by d_mul[p] while it is loaded (d_mul: floats in device memory,
 * kept by the caller, indexed by position).  zoom's x stage reads ONE array twice this way -- T[u] cos(theta u) in its place and
 * T[M - x] sin(theta (M - x)) mirrored -- instead of having a kernel write both products out.  Returns 1 when honoured (f32 plans whose
 * first pass is a listed specialised row or column REDFT01 pass along `axis` with a window), 0 when not (nothing changes); d_mul = NULL: off. */
int dspfft_plan_set_input_modulation(dspfft_plan plan, int axis, const float *d_mul, int reversed_from);
/* Optional: output sample j of `axis` is multiplied by (-1)^j, fused into that axis's pass (with REDFT01 on index-reversed input this is
 * the sine counterpart of the transform: sum_u D[u] sin(pi (j + 1/2) u / M) = (-1)^j 1/2 REDFT01(E)[j], E[u'] = D[M - u'] -- the second
 * half of zoom's shifted cosine series).  Returns 1 when honoured (f32 plans whose pass along `axis` is a listed specialised column or row REDFT01
 * pass and the last of the plan), 0 when not (nothing changes).  Combined with dspfft_execute_masked_accumulate(plan, in, work, acc, NULL,
 * ...) the signed result is added into `acc` by the same pass. */
int dspfft_plan_set_output_alternate(dspfft_plan plan, int axis, int on);
/* d_out = a(d_in_a) + b(d_in_b): the sum of two plans' results, each with its own scale, input window and output sign.  ONE launch
 * that writes d_out once when both are f32 one-axis plans on the same listed row REDFT01 kernel over the same output lines (zoom's
 * x stage by fast transforms: cosine part + sine part); otherwise dspfft_execute(a) followed by an accumulating execution of b, for
 * which b must be a one-pass plan (-2 otherwise).  d_out must not overlap the inputs. */
int dspfft_execute_sum2(dspfft_plan a, dspfft_plan b, const float *d_in_a, const float *d_in_b, float *d_out, void *hip_stream);

/* ---- rows of a phase-shifted cosine series (zoom's x stage: zoom/zoom.c:361-368 with the basis of :36-68 on a DCT-III grid) ----
 *     out[j][b][c] = scale * sum'_{u < cw} in[j][u][c] cos(u (pi (b + 1/2) / M + theta)),    b < vw <= M, c < 3, j < lines
 * (sum' halves u = 0: zoom.c:364 `coeffs[..]/2`).  With theta = 0 this is 1/2 REDFT01_M of the zero-padded line; a non-zero theta is a
 * pan by theta M / pi samples (zoom.c:49-57 `offset`), which two transforms would need (cosine and sine part: dspfft_execute_sum2).
 * Here both parts ride through ONE half-length FFT per channel as the halves of a packed-FP32 pair: a kernel of its own
 * (dspfun_amd/csrc/dct_duo.h), three barrier-separated phases per channel, whole pixels stored once.
 * Scaled lengths M with a listed kernel only (spec_list.h DSPFFT_ZOOMX_SPECS: 7680, 5760, 3840, 2560, 1920, 1280): create returns -2 for the
 * others and the caller keeps dspfft_execute_sum2.  Lines: `in` cw pixels of 3 floats, in_pitch floats apart; `out` vw pixels, out_pitch
 * floats apart; 4-byte alignment suffices.  theta and scale are arguments of the execution (a pan re-plans nothing): the multiplier table
 * they determine (M/2 .. 2M slots of 16 bytes, evaluated in double on the device) is rebuilt per call, on the stream, in the plan's
 * own buffer -- one execution of a plan at a time. */
typedef struct dspfft_cosrows_s *dspfft_cosrows;
int dspfft_cosrows_create(dspfft_cosrows *plan, int M, int cw, int vw, int lines);
int dspfft_cosrows_execute(dspfft_cosrows plan, const float *d_in, long long in_pitch, float *d_out, long long out_pitch, double theta, double scale, void *hip_stream);
void dspfft_cosrows_destroy(dspfft_cosrows plan);

/* ---- rows of a cosine series at ANY sample spacing, by the chirp-z transform (zoom's product at scales off the DCT-III grid and for
 * the `centered` basis: zoom/zoom.c:36-68,361-375; dspfun_amd/csrc/dct_czt.h) ----
 *     out[b] = scale * sum'_{n < nc} in[n] cos(n (omega b + phi)),    b < nout           (sum' halves n = 0)
 * per line: a circular convolution of a listed smooth length P >= nc + nout - 1 (1200 ... 19200: spec_list.h DSPFFT_CZT_SPECS) in LDS,
 * forward FFT, times the chirp's spectrum, inverse FFT.  create returns -2 when nc + nout - 1 exceeds the longest listed P.
 * Lines come in `groups` of `group` members (1, or 3 = the channels of an image row, which share their output cache lines and are then run
 * on one XCD back to back): line (g, m) reads in[g * in_group + m * in_pitch + n * es_in] and writes out[g * out_group + m * out_pitch + b * es_out].
 * omega, phi and scale are arguments of the execution; the tables they determine live in the plan (one execution of a plan at a time);
 * the chirp's spectrum is rebuilt only when omega changes. */
typedef struct dspfft_cztrows_s *dspfft_cztrows;
int dspfft_cztrows_create(dspfft_cztrows *plan, int nc, int nout, int lines, int group);
int dspfft_cztrows_execute(dspfft_cztrows plan, const float *d_in, long long in_group, long long in_pitch, int es_in,
                           float *d_out, long long out_group, long long out_pitch, int es_out, double omega, double phi, double scale, void *hip_stream);
int dspfft_cztrows_length(dspfft_cztrows plan);      /* the convolution length P the plan runs on */
void dspfft_cztrows_destroy(dspfft_cztrows plan);
/* out[c * out_pitch + r] = in[r * in_pitch + c], r < rows, c < cols (the re-layout between the two axes of a chirp-z zoom frame) */
int dspfft_transpose_f32(float *d_out, long long out_pitch, const float *d_in, long long in_pitch, int rows, int cols, void *hip_stream);
/* Optional, for owner ids that stay the same over the frames of a scan (every method but box, whose ids are stamped per frame):
 * records the (min, max) owner id of every column tile of `plan`, so that a later dspfft_execute_masked_accumulate with the SAME
 * d_ids pointer and elems_per_id leaves a tile alone -- without reading its owner ids -- when `id` lies outside its range.  Call it
 * again after rewriting the ids; d_ids = NULL forgets.  Results are the same with or without it.
 * For single-precision plans it also copies the ids into the order the column tiles read them, one element one id, one byte each (two when
 * some id is 255 or more; none when some id is 65535 or more): plan-owned device memory of 1 or 2 bytes per element of the image.  The
 * masked column pass of the step then reads that table instead of d_ids (8K RGB frame step 0.429 -> 0.379 ms); it synchronises
 * hip_stream once to look at the ranges.
 * The library cannot see writes to the id array: rewriting it (dspfft_scan_stamp, dspfft_scan_frame_ids, dspfft_scan_index_to_frame_ids
 * or your own kernel on the same buffer) WITHOUT preparing again makes later steps skip the wrong tiles.  The prepared ranges and the
 * per-tile flags are scratch of the plan: masked executions of one plan must be serialised on one stream (one plan per stream otherwise). */
int dspfft_plan_scan_prepare(dspfft_plan plan, const uint32_t *d_ids, int elems_per_id, void *hip_stream);
int dspfft_execute_masked_accumulate_f64(dspfft_plan plan, const double *d_in, double *d_work, double *d_acc,
                                         const uint32_t *d_ids, uint32_t id, int elems_per_id, void *hip_stream);

/* The shape of FFTW's guru interface (fftw_plan_guru_r2r; not called by the reference's tools, which loop over blocks
 * on the host instead: motion/motion.c:591-615): every transformed dimension and every batch dimension carries its own
 * extent and input / output strides (in elements).  One plan then covers, e.g., all 8x8x8 blocks of a [D][H][W] volume
 * (motion --blocksize 8x8x8, motion/README.md): dims = {8,HW,HW},{8,W,W},{8,1,1}, howmany_dims =
 * {D/8,8HW,8HW},{H/8,8W,8W},{W/8,8,8}.  rank 1..3, howmany_rank 0..6; f64 != 0 selects double samples.  Lengths up to
 * 32 run one line per thread in registers. */
typedef struct { int n; int is; int os; } dspfft_iodim;
/* Planning effort for the plans created after the call, on ANY thread of the process (plans already made keep theirs); a thread may override it for
 * the plans it makes itself with dspfft_set_thread_plan_effort (effort < 0 removes the override; dspfft_get_thread_plan_effort: -1 when none), which is
 * what the FFTW shim does around each fftw(plan_many_r2r).  dspfft_get_plan_effort returns the value in force for the calling thread.
 * 0 (default): a frame size without a compile-time-specialised kernel (spec_list.h) runs on the runtime-geometry kernels.
 * > 0: such sizes get RowSpecT / ColSpecT kernels COMPILED AT PLAN TIME (hiprtc: about a second per new size, then cached in
 * $DSPFFT_JIT_CACHE or ~/.cache/dspfft-jit) and run 1.3-1.8x faster from then on.
 * >= 2: several candidates (radix orders, thread counts, column tile widths) are compiled and TIMED on a scratch buffer, the fastest
 * is kept (the caller's arrays are never touched).  The FFTW shim maps FFTW_ESTIMATE to 0, FFTW_MEASURE (scan.c:359) to 1 and
 * FFTW_PATIENT / FFTW_EXHAUSTIVE (motion.c:93-103) to 2.  DSPFFT_JIT=1 / 2 in the environment forces compilation on / off,
 * DSPFFT_JIT_TUNE=1 / 2 the timed search. */
void dspfft_set_plan_effort(int effort);
int dspfft_get_plan_effort(void);
void dspfft_set_thread_plan_effort(int effort);
int dspfft_get_thread_plan_effort(void);
/* Diagnostic for the FFTW shim (include/fftw3.h): how many fftw(execute) calls of this process carried their input up as packed non-zero 4 KB blocks
 * instead of one dense copy.  Arrays of 32 MB and more are read by host threads first (DSPFFT_UPLOAD_THREADS, default = the CPU quota, at most 12;
 * 0 = never); an array more than a quarter non-zero goes up dense and is looked at again only after 1, 2, 4 ... 64 further executes.  What arrives on the
 * device is the same bytes either way (scan/scan.c:429-447: <= 3 % of `reconstruction` is set per output frame). */
unsigned long long dspfft_fftw_sparse_uploads(void);
int dspfft_plan_guru_r2r(dspfft_plan *plan, int rank, const dspfft_iodim *dims, int howmany_rank, const dspfft_iodim *howmany_dims,
                         const int *kinds, int f64);

/* As dspfft_plan_many_r2r, with the order of the axis passes chosen: first_axis_first = 0 is the default (last axis
 * first), 1 runs axis 0 first and the contiguous axis last.  The results are the same; the order decides which pass
 * reads the caller's input layout and which one ends the plan. */
int dspfft_plan_many_r2r_ordered(dspfft_plan *plan, int rank, const int *n, int howmany,
                                 const int *inembed, int istride, int idist,
                                 const int *onembed, int ostride, int odist, const int *kinds, int first_axis_first);

/* motion's transform -> filter -> inverse loop (motion/motion.c:641-753) as one call.  `fwd` is a REDFT10 plan in
 * the default order, `inv` a REDFT01 plan created with first_axis_first = 1, in place on fwd's output layout; then
 * fwd's last pass and inv's first pass run along the same axis, and when both have a specialised column kernel
 * they execute as ONE launch: the tile stays in LDS through forward transform, filter and inverse transform
 * (8 B/sample of HBM traffic instead of 24 for that axis and the filter).  Otherwise -- also when a plan carries
 * an input window, modulation or alternating output sign on that axis (dspfft_plan_set_*: the fused kernel and the 8-bit
 * row ends are plain instantiations) -- the three steps run separately.  filter == NULL skips the filter.  The filter is motion.c:683-744 (see dspfft_motion_filter below):
 * positions are taken inside blocks of block_depth planes of minbuf_hw[0] x minbuf_hw[1] elements (block_depth =
 * the embedding depth for one 3-D block, 1 when every frame is its own block, motion's default -b 0x0x1);
 * d_coeffs_coded (device, may be NULL) is incremented by the number of non-zero quantised coefficients.
 * Small blocks (motion --blocksize 8x8x8 and the like: every extent 4, 8 or 16, 2-D blocks up to 32; f32; x contiguous; planned
 * with dspfft_plan_many_r2r as a block-major stack or with dspfft_plan_guru_r2r where they lie in a volume): dspfft_execute runs all
 * axes of a block in ONE pass, and this call -- the _u8 form included -- runs load, forward transform, filter (positions are the
 * block's own coordinates), inverse transform and store as ONE kernel; the inverse plan needs no particular pass order then.
 * Buffers must be 16-byte aligned (8-bit ones: 4-byte) for that path; a filtered roundtrip over the blocks of a volume has no
 * other path and fails otherwise. */
typedef struct {
	int active[3];            /* block extent {d, h, w} actually transformed */
	int minbuf_hw[2];         /* rows and row pitch of the buffer's planes */
	int block_depth;
	int band_begin[3], band_end[3];
	float damp, boost, threshold_lo, threshold_hi;
	int preserve_dc;          /* 0 none, 1 dc, 2 grey */
	float grey_add, quantizer;
} dspfft_motion_filter_params;
int dspfft_execute_roundtrip(dspfft_plan fwd, dspfft_plan inv, const float *d_in, float *d_out,
                             const dspfft_motion_filter_params *filter, unsigned long long *d_coeffs_coded, void *hip_stream);

/* `scaled != block` (motion/motion.c:535-552,566,644-647,748-759): `inv` may be planned over DIFFERENT extents than `fwd` inside the
 * same embedding (same strides; howmany = 1).  Larger extents zero-pad the spectrum (band-limited upscale), smaller ones truncate
 * it (downscale); the filter's `active` is min(block, scaled) per axis.  The transforms then run unfused.  The work buffer is
 * zeroed first (motion.c:619) except in place on a float buffer, where the caller has zeroed everything outside the block. */

/* The same with motion's 8-bit samples at both ends (motion/motion.c:617-640 load, :760-776 store): d_in and d_out
 * hold uint8 samples in the plans' input / output element layout, d_work is a float buffer in the plans' working
 * layout.  When the first forward pass and the last inverse pass are planar specialised row passes they read the
 * bytes and write quantise(value * out_mul) themselves (5 B/sample each instead of 13); otherwise the conversions
 * run as separate sweeps (dspfft_u8_to_f32 / dspfft_f32_to_u8), which requires input, work and output layouts to be
 * identical and dense. */
int dspfft_execute_roundtrip_u8(dspfft_plan fwd, dspfft_plan inv, const uint8_t *d_in, uint8_t *d_out, float *d_work, double out_mul,
                                const dspfft_motion_filter_params *filter, unsigned long long *d_coeffs_coded, void *hip_stream);

/* Replaces fftw(destroy_plan). */
void dspfft_destroy_plan(dspfft_plan plan);

/* Human-readable list of the passes a plan runs (kernel shape, radices, tile, LDS bytes). */
int dspfft_plan_describe(dspfft_plan plan, char *buf, size_t buflen);

/* Algorithmic bytes one execute moves (read once + write once per transformed sample, 8 B/sample
 * for f32, 16 for f64) -- the numerator of bench.py's roofline figure. */
size_t dspfft_plan_algorithmic_bytes(dspfft_plan plan);

const char *dspfft_last_error(void);
const char *dspfft_version(void);

/* ---- device-side helpers around the transform ---- */

/* scan/scan_methods.c:77-115 (scan_zigzag): lin[i - first] = y*w + x of scan index i, i in [first, first+count).
 * d_lin: device uint32 array.  Integer, bit-exact. */
int dspfft_scan_zigzag(uint32_t *d_lin, uint32_t w, uint32_t h, uint64_t first, uint64_t count, void *hip_stream);

/* scan/scan.c:429-432,445: recon = 0 everywhere except the `count` listed pixels (all c channels copied
 * from coeffs), DC pixel cleared. */
int dspfft_scan_scatter(float *d_recon, const float *d_coeffs, const uint32_t *d_lin, uint64_t count,
                        uint64_t npixels, int channels, void *hip_stream);

/* The whole per-frame step of scan's hot loop (scan/scan.c:429-432,445-459) as one fused execution of an
 * inverse plan:   d_acc += plan( d_in restricted to elements e with d_ids[e / elems_per_id] == id )
 * The first pass masks while loading (no zeroed `reconstruction` buffer, no scatter), the last pass
 * adds into d_acc while storing (no `image` buffer, no separate sum pass): 21.3 B/sample of traffic for
 * a 3-channel image instead of 32.  d_work: scratch with the plan's output layout.  d_ids == NULL
 * disables masking.  elems_per_id = channels for the image tools' interleaved buffers.
 * Plans with split column passes (8K-class frames) skip the column tiles none of whose coefficients belongs to `id` and keep
 * per-tile flags for the row pass in scratch owned by the plan: run one fused step per plan at a time (concurrent streams take
 * one plan each); d_work then holds garbage in the skipped tiles, as scratch may. */
int dspfft_execute_masked_accumulate(dspfft_plan plan, const float *d_in, float *d_work, float *d_acc,
                                     const uint32_t *d_ids, uint32_t id, int elems_per_id, void *hip_stream);

/* d_ids[y*w+x] = (zigzag scan index of (y,x)) / step -- the output frame (scan/scan.c:421-427) that
 * reconstructs coefficient (y,x); the DC pixel gets 0xFFFFFFFF (it is pre-added, scan.c:377-383,445). */
int dspfft_scan_zigzag_frame_ids(uint32_t *d_ids, uint32_t w, uint32_t h, uint64_t step, void *hip_stream);

/* ---- the other scan methods on the device (scan/scan_methods.c:59-67,122-208,240-331), SURVEY.md 8f #3 ----
 * Method numbers follow host/scan_orders.h.  Every generator is bit-exact against the reference's (integer arithmetic; radial's
 * rint(hypot) is evaluated exactly).  How a frame loop uses them:
 *   all methods but box: dspfft_scan_frame_ids once, then dspfft_execute_masked_accumulate(..., frame, ...) per output frame;
 *   box (a pixel can belong to several scan indices, and its first leg runs out of the image on tall frames): per frame
 *     dspfft_scan_coords of the frame's scan indices -> dspfft_scan_stamp(ids, ..., frame) -> the same fused execution;
 *   magnitude (needs the coefficients): dspfft_scan_magnitude_index -> dspfft_scan_index_to_frame_ids;
 *   file: parsed on the host (host/scan_orders.c scan_order_read_file), uploaded as an index array or as coordinate lists. */
enum {
	DSPFFT_SCAN_HORIZONTAL = 0, DSPFFT_SCAN_VERTICAL = 1, DSPFFT_SCAN_ZIGZAG = 2, DSPFFT_SCAN_ROW = 3, DSPFFT_SCAN_COLUMN = 4,
	DSPFFT_SCAN_DIAGONAL = 5, DSPFFT_SCAN_MIRROR = 6, DSPFFT_SCAN_BOX = 7, DSPFFT_SCAN_IBOX = 8, DSPFFT_SCAN_RADIAL = 9, DSPFFT_SCAN_IRADIAL = 10
};
/* number of scan indices (scan_context.c:30 via the method's limit function) and the most coordinates one index yields (host arithmetic) */
uint64_t dspfft_scan_limit(int method, uint32_t w, uint32_t h);
uint64_t dspfft_scan_max_interval(int method, uint32_t w, uint32_t h);
/* entries per scan index in the coordinate lists below: max_interval, + 1 for box / ibox (the entry scan.c:346 over-allocates; ibox's
 * index 0 emits w + h coordinates against max_interval = w + h - 1, scan_methods.c:135-144,502) */
uint64_t dspfft_scan_coord_slots(int method, uint32_t w, uint32_t h);
/* d_index[y*w+x] = the scan index that yields pixel (y,x); every method except box (no single owner) */
int dspfft_scan_owner_index(uint32_t *d_index, int method, uint32_t w, uint32_t h, void *hip_stream);
/* d_ids[y*w+x] = owner index / step, the DC pixel 0xFFFFFFFF: the generalisation of dspfft_scan_zigzag_frame_ids (which it calls for zigzag) */
int dspfft_scan_frame_ids(uint32_t *d_ids, int method, uint32_t w, uint32_t h, uint64_t step, void *hip_stream);
/* coordinate lists: for scan indices [first, first+count), slot j < dspfft_scan_coord_slots of index i holds y*w+x of its j-th coordinate,
 * or 0xFFFFFFFF (slot beyond the index's interval, or a box coordinate past the end of the image); all methods but radial / iradial.
 * d_lin holds count * dspfft_scan_coord_slots entries. */
int dspfft_scan_coords(uint32_t *d_lin, int method, uint32_t w, uint32_t h, uint64_t first, uint64_t count, void *hip_stream);
/* d_ids[d_lin[t]] = frame_id for every valid entry (the DC pixel is left alone): builds the mask of ONE frame from coordinate lists;
 * ids of earlier frames need no clearing as long as frame ids are not reused (initialise d_ids to 0xFFFFFFFF once) */
int dspfft_scan_stamp(uint32_t *d_ids, const uint32_t *d_lin, uint64_t nslots, uint32_t frame_id, void *hip_stream);
/* in place: ids[p] = index[p] / step, DC pixel 0xFFFFFFFF (for index arrays from dspfft_scan_magnitude_index or a `file` order) */
int dspfft_scan_index_to_frame_ids(uint32_t *d_ids, uint64_t npixels, uint64_t step, void *hip_stream);
/* scan_methods.c:240-296: d_index[p] = scan index of pixel p when pixels are ordered by descending magnitude key
 * (sum_z |c|, x sqrt2 per non-zero index, optionally quantised by qfactor; 0 = off), INCLUDING the reference's grouping rule (an
 * element whose key differs from its predecessor's still joins the predecessor's index; the next element opens a new one).
 * Ties: the reference leaves their order to qsort; here equal keys keep raster order (stable sort) -- documented, deterministic.
 * Synchronises the stream; *limit receives the number of scan indices.  d_work: dspfft_scan_magnitude_work_bytes(w, h) bytes. */
size_t dspfft_scan_magnitude_work_bytes(uint32_t w, uint32_t h);
int dspfft_scan_magnitude_index(uint32_t *d_index, const float *d_coeffs, uint32_t w, uint32_t h, int channels, double qfactor,
                                void *d_work, size_t work_bytes, uint32_t *limit, void *hip_stream);

/* ---- the rest of motion's block loop (motion/motion.c), HIP-only entry points ---- */
enum { DSPFFT_MOTION_NONE = 0, DSPFFT_MOTION_ABS = 1, DSPFFT_MOTION_SHIFT = 2, DSPFFT_MOTION_FLAT = 3, DSPFFT_MOTION_COPY = 4 };
/* motion.c:617-640: the {n[0],n[1],n[2]} corner of an 8-bit buffer -> float, both laid out as planes of minbuf_hw[0] x minbuf_hw[1];
 * ispec_mode decodes an input spectrogram (--ispec: shift uses ic = 127.5 / log1p(N normalization 255 8), :568,626-628) */
int dspfft_motion_load_u8(float *d_coeffs, const uint8_t *d_pix, const int n[3], const int minbuf_hw[2], int ispec_mode, double ic, double normalization, void *hip_stream);
/* motion.c:756-776: pel = coeff * scalefactor * normalization, then the --spec encode (abs / shift with constant c, :755,762-763; flat :764)
 * or * normalization again (none / copy, :767), clamp and lround to 8 bits */
int dspfft_motion_store_u8(uint8_t *d_pix, const float *d_coeffs, const int n[3], const int minbuf_hw[2], int spec_mode,
                           double scalefactor, double normalization, double c, void *hip_stream);
/* the same two with float pixels (motion's float_pixels: planar float formats): the load reads sample * 255 (motion.c:623), the store writes
 * pel / 255 unclamped (:774) */
int dspfft_motion_load_f32(float *d_coeffs, const float *d_pix, const int n[3], const int minbuf_hw[2], int ispec_mode, double ic, double normalization, void *hip_stream);
int dspfft_motion_store_f32(float *d_pix, const float *d_coeffs, const int n[3], const int minbuf_hw[2], int spec_mode,
                            double scalefactor, double normalization, double c, void *hip_stream);
/* motion.c:652-668 (--coeff-limit): keep the `keep` coefficients of largest magnitude among d_coeffs[0 .. count), zero the rest.
 * Radix select on the device (four histogram passes over the bits of |c|, no sort).  Ties at the threshold: the reference's choice
 * depends on qsort; here the earliest in buffer order are kept -- documented, deterministic. */
size_t dspfft_motion_topn_work_bytes(size_t count);
int dspfft_motion_topn(float *d_coeffs, size_t count, size_t keep, void *d_work, size_t work_bytes, void *hip_stream);
const char *dspfft_motion_last_error(void);

/* scan/scan.c:451-459 arithmetic: sum += image (len floats). */
int dspfft_accumulate(float *d_sum, const float *d_image, uint64_t len, void *hip_stream);

/* scan/scan.c:377-383: sum[p*channels + z] = coeffs[z] for every pixel p. */
int dspfft_broadcast_dc(float *d_sum, const float *d_coeffs, uint64_t npixels, int channels, void *hip_stream);

/* motion/motion.c:617-638 (ispec none, !linear, 8-bit input): coeffs[i] = (float)pix[i]. */
int dspfft_u8_to_f32(float *d_dst, const uint8_t *d_src, uint64_t len, void *hip_stream);

/* motion/motion.c:756-776 (spec none, !linear, 8-bit output): pix = clamp(lround(c * mul), 0, 255). */
int dspfft_f32_to_u8(uint8_t *d_dst, const float *d_src, double mul, uint64_t len, void *hip_stream);

/* ---- zoom's dense basis product on the f32 matrix cores ---- */

/* zoom/zoom.c:37-41: number of cosine components kept for a (scale, length) pair. */
size_t dspfft_zoom_ncomponents(double scale_num, double scale_den, size_t len);

/* zoom/zoom.c:36-68 (generate_scaled_basis): d_basis[b*nc + n], nc = dspfft_zoom_ncomponents(...), with
 * column 0 = 1/2 (the halved DC term of zoom.c:364,369) and column n >= 1 = cos(pi (k+1/2) n / N).
 * type: 0 interpolated, 1 centered, 2 native (zoom/zoom.c:20-26). */
int dspfft_zoom_basis(float *d_basis, int type, double scale_num, double scale_den, double offset,
                      size_t nvectors, size_t len, void *hip_stream);

/* zoom/zoom.c:361-375: out (vh x vw x 3) = YB . (C_z[:ch,:cw] . XB^T) / (w h) for the three channels of the
 * unnormalised REDFT10^2 coefficients d_coeffs (h x w x 3 interleaved).  d_work: dspfft_zoom_work_floats(). */
size_t dspfft_zoom_work_floats(int w, int h, size_t ch, int vw);
int dspfft_zoom_product(const float *d_coeffs, int w, int h, const float *d_xb, size_t cw, const float *d_yb, size_t ch,
                        float *d_out, int vw, int vh, float *d_work, void *hip_stream);

/* The GEMM underneath (also the building block for applybasis' basis x pixel sums):
 * C[m*ldc + n*cs] = alpha * sum_k A[m*lda + k] * B[n*ldb + k], `batch` problems at offsets sa/sb/sc. */
int dspfft_gemm_nt_f32(const float *A, const float *B, float *C, int M, int N, int K,
                       long long lda, long long ldb, long long ldc, int cs,
                       int batch, long long sa, long long sb, long long sc, float alpha, void *hip_stream);
const char *dspfft_zoom_last_error(void);

/* The same frame by fast transforms, for the scales at which an axis's output samples lie on a DCT-III grid: `interpolated` (0) or
 * `native` (2) basis with len * num / den an integer M on both axes and a viewport of at most Mx x My samples (any offset: a pan
 * re-plans nothing).  Per axis  out[b] = 1/2 REDFT01_M(C[n] cos(theta n))[b] - (-1)^b 1/2 REDFT01_M(C[M - n'] sin(theta (M - n')))[b],
 * theta = pi (offset + (s - 1) / 2) / M (interpolated), pi offset / M (native)  (zoom/zoom.c:49-57): two
 * length-M REDFT01 executions per axis on the row / column kernels instead of the dense product -- BASELINE config 3 (4x of
 * 1920x1080): 310 GFLOP become about 1 GB of streaming in three transform launches.  dspfft_zoomfft_create returns -2 when the scale,
 * basis or viewport does not qualify (use dspfft_zoom_product then); d_coeffs, d_out (vh x vw x 3) and d_work
 * (dspfft_zoomfft_work_floats floats) 16-byte aligned.  One object serves one stream at a time. */
typedef struct dspfft_zoomfft_s *dspfft_zoomfft;
int dspfft_zoomfft_create(dspfft_zoomfft *z, int w, int h, int type, double xscale_num, double xscale_den, double yscale_num, double yscale_den,
                          int vw, int vh);
size_t dspfft_zoomfft_work_floats(dspfft_zoomfft z);
int dspfft_zoomfft_execute(dspfft_zoomfft z, const float *d_coeffs, double vx, double vy, float *d_out, float *d_work, void *hip_stream);
void dspfft_zoomfft_destroy(dspfft_zoomfft z);
const char *dspfft_zoomfft_last_error(void);
/* zoom's frame at ANY scale, offset and basis (interpolated, centered, native) by chirp-z transforms along both axes: same arguments as
 * dspfft_zoomfft_*, no restriction on the scaled lengths (dspfft_zoomfft_* needs them integer and refuses `centered`).  -2 when an axis is
 * longer than the longest listed convolution (coefficients + samples - 1 > 19200): the dense product (dspfft_zoom_product) remains. */
typedef struct dspfft_zoomczt_s *dspfft_zoomczt;
int dspfft_zoomczt_create(dspfft_zoomczt *z, int w, int h, int type, double xnum, double xden, double ynum, double yden, int vw, int vh);
size_t dspfft_zoomczt_work_floats(dspfft_zoomczt z);
int dspfft_zoomczt_execute(dspfft_zoomczt z, const float *d_coeffs, double vx, double vy, float *d_out, float *d_work, void *hip_stream);
void dspfft_zoomczt_destroy(dspfft_zoomczt z);

/* ---- applybasis' basis x pixel partial sums on the matrix cores ----
 * applybasis/applybasis.c:410-431, forward direction:
 *   out[k_h][k_w][n_h][n_w][j] = sum_{s_h < Ph, s_w < Pw} f(k_h + offh, n_h Ph + s_h, h) f(k_w + offw, n_w Pw + s_w, w) pix[..][j]
 * func: 0 dft, 1 idft, 2..5 dct1-4, 6..9 dst1-4, 10 wht, 11 dht (applybasis.c:77-140); d_pixels: h x w x 3 f32 already
 * range-mapped (applybasis.c:358-360); d_out: Kh*Kw*(h/Ph)*(w/Pw)*3 complex (re, im) floats in the order of the tool's
 * `.coeff` dump (applybasis.c:443).  d_work: dspfft_applybasis_work_floats() floats. */
size_t dspfft_applybasis_work_floats(int w, int h, int Kw, int Kh, int Pw, int Ph, int func);
int dspfft_applybasis_partsums(float *d_out, const float *d_pixels, int w, int h, int func, int ortho,
                               int Kw, int Kh, int Pw, int Ph, long long offw, long long offh,
                               float *d_work, void *hip_stream);

/* The general form: N partial-sum blocks of P pixels per axis chosen freely (N x P <= image), real or complex pixels.
 *   forward  (applybasis.c:378-389 with !inverse):  K = terms,       N = image size / partsum
 *   --inverse                                    :  K = image size,  N = terms / partsum
 *   .coeff input (applybasis.c:319-338): complex pixels -- d_pix_im non-NULL (planes h x w x 3, like d_pix_re)
 * --offset (applybasis.c:416-420 adds it to `bi`): inverse = 0 -> f(k + off, n P + s); inverse = 1 (`n = &bi`, :372-378) ->
 * f(k, (n + off) P + s) while the pixel read stays at n P + s.
 * Each call is two batched GEMM launches (three basis / layout kernels around them), whatever the function. */
size_t dspfft_applybasis_work_floats_ex(int w, int h, int Kw, int Kh, int Nw, int Nh, int func);
int dspfft_applybasis_partsums_ex(float *d_out, const float *d_pix_re, const float *d_pix_im, int w, int h, int func, int ortho,
                                  int Kw, int Kh, int Nw, int Nh, int Pw, int Ph, long long offw, long long offh, int inverse,
                                  float *d_work, void *hip_stream);
/* applybasis.c:392-442: the rendered frame from the partial sums.  d_frame: fh x fw x 4 floats (RGBA) with
 * fw = Kw Nw scale + padding T_w + padding (T = K forward, N with --inverse), same for fh; filled with padcolor first.
 * plane: 0 real 1 imaginary 2 magnitude 3 phase (:20-31); rescale0 / rescale1: 0 linear 1 log 2 gain 3 level, rescale1 = -1 for a
 * single type, else the two are interpolated as :433-438 does; range: 0 shift / shift2, 1 abs, 2 invert, 3 hue (:46-76);
 * coeff_scale as :399-407; insize_wh = input width x height. */
int dspfft_applybasis_render(float *d_frame, const float *d_partsums, int Kw, int Kh, int Nw, int Nh, int inverse, int scale, int padding,
                             int plane, int rescale0, int rescale1, int range, double coeff_scale, double insize_wh,
                             const float padcolor[4], void *hip_stream);

/* ---- elementwise stages either side of the transform, on the device ---- */

/* spec/spec.c:81-139 on uniform-range coefficients (after the fused normalisation): f *= gain; divisor per channel
 * from rangetype (0 one, 1 dc, 2 dcs; spec.c:92-110); scaletype 0 log (copysign(log1p|f|)/log1p(max), :113-118) or
 * 1 linear (:120-122); signtype 0 abs, 1 shift, 2 saturate, 3 retain (:124-139).  `gain` is the resolved multiplier
 * (native 127.5*sqrt(4wh), reference 127.5*1024, or custom; spec.c:82-87). */
int dspfft_spec_encode(float *d_f, size_t npixels, int channels, double gain, int rangetype, int scaletype, int signtype, void *hip_stream);

/* spec/ispec.c:100-151, the inverse (the separate sign-map image of :91-99 is not handled).  dc: `channels` doubles
 * from the spectrogram's "DC" property (host memory; needed for rangetype dc/dcs and for restore_dc, ispec.c:161-163). */
int dspfft_ispec_decode(float *d_f, size_t npixels, int channels, double gain, int rangetype, int scaletype, int signtype,
                        const double *dc, int restore_dc, void *hip_stream);
/* spec/ispec.c:91-99, run BEFORE dspfft_ispec_decode of an `abs` spectrogram that comes with a sign-map image (-m): every sample but
 * the first pixel's takes the sign of (map - 128); the first pixel of the map carries the DC terms instead (DC[z] = map[z] / 255,
 * host arithmetic on channels bytes -- pass them to dspfft_ispec_decode as `dc`). */
int dspfft_ispec_signmap(float *d_f, const uint8_t *d_map, size_t npixels, int channels, void *hip_stream);

/* motion/motion.c:683-744 on one block of uniform-range coefficients embedded in {., minbuf_h, minbuf_w}: damp outside /
 * boost inside the band-pass box [band_begin, band_end), threshold on |c| (threshold_hi <= 0 disables; bounds already scaled
 * as motion.c:571-572), DC preservation (0 none, 1 dc, 2 grey with grey_add = (1 - (dcstop ? damp : boost)) * 127.5 /
 * (normalization^2 * scalefactor), :736), quantisation round(c/Q)*Q (Q <= 0 disables).  d_coeffs_coded (optional, device)
 * accumulates the count of non-zero quantised coefficients (:744).  All index triples are {d, h, w}. */
int dspfft_motion_filter(float *d_coeffs, const int active[3], const int minbuf_hw[2], const int band_begin[3], const int band_end[3],
                         float damp, float boost, float threshold_lo, float threshold_hi, int preserve_dc, float grey_add,
                         float quantizer, unsigned long long *d_coeffs_coded, void *hip_stream);
/* scan/scan.c:20-41,449 (pruned_idct) fused with the accumulate of :451-459:
 *   sum[y][x][z] += sum_n coeffs[lin[n]][z] * B_h[y][cy_n] * B_w[x][cx_n],  B_N[k][j] = j ? 2 cos(pi j (k+1/2)/N) : 1
 * for the `ncoords` coefficients (device list of y*w+x offsets) one output frame adds.  The reference takes this path
 * when step * max_interval <= log2(w*h) (scan.c:349-350).  The caller leaves the DC pixel out of the list (scan.c:445). */
int dspfft_scan_pruned_accumulate(float *d_sum, const float *d_coeffs, const uint32_t *d_lin, int ncoords,
                                  int w, int h, int channels, void *hip_stream);
/* the same with its basis table in a caller-provided buffer of dspfft_scan_pruned_work_floats(ncoords, w, h) floats: no allocation
 * inside the call, so a scan loop on the pruned path can be captured into a hipGraph (dspfft_scan_pruned_accumulate allocates per call) */
size_t dspfft_scan_pruned_work_floats(int ncoords, int w, int h);
int dspfft_scan_pruned_accumulate_ws(float *d_sum, const float *d_coeffs, const uint32_t *d_lin, int ncoords,
                                     int w, int h, int channels, float *d_work, void *hip_stream);
const char *dspfft_pointwise_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
