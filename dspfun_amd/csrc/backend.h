// backend.h -- what the plan/engine layer needs from the device side.  The product links
// backend_hip.hip (HIP kernels on gfx950).  tests/emul links a CPU emulation of the SAME phase
// functions for CPU-only unit tests of the planner and kernel logic; that build is never shipped.
#pragma once
#include <stddef.h>
#include <stdlib.h>
#include "dct_core.h"
#include "motion_filter.h"
#include "dct_spec.h"
#include "dct_duo.h"
#include "dct_czt.h"
#include "scan_core.h"
#include "block_core.h"

namespace dspfft {

struct LaunchGeom {
	int nwg;          // workgroups
	int nthr;         // threads per workgroup
	size_t lds_bytes; // dynamic LDS
};

void *be_alloc(size_t bytes);
void be_free(void *p);
int be_upload(void *dst, const void *src, size_t bytes);   // host -> device table upload (synchronous)
size_t be_max_lds();                                        // usable LDS bytes per workgroup
const char *be_name();

int be_launch_row(const PassArgs &a, const LaunchGeom &g, void *stream);
int be_launch_col(const PassArgs &a, const LaunchGeom &g, void *stream);
int be_launch_dense(const DenseArgs &a, const LaunchGeom &g, void *stream);
// double-precision samples (the fftw_ API of spec's default build): generic kernels only
int be_launch_row(const PassArgsD &a, const LaunchGeom &g, void *stream);
int be_launch_col(const PassArgsD &a, const LaunchGeom &g, void *stream);
int be_launch_dense(const DenseArgsD &a, const LaunchGeom &g, void *stream);
// lengths 1..32: one line per thread, in registers
int be_launch_tiny(const TinyArgs &a, void *stream);
int be_launch_tiny(const TinyArgsD &a, void *stream);
// COL pass with the tile's DFT done by Bluestein's convolution (lengths with prime factors > 13)
int be_launch_blue(const BlueArgs &a, const LaunchGeom &g, void *stream);
int be_launch_blue(const BlueArgsD &a, const LaunchGeom &g, void *stream);

// Plan-time compilation (hiprtc) of jit_kernels.h for a RowSpecT / ColSpecT instance named by its C++ type; incdir = where the
// library's headers lie.  funcs[0 / 1] = the REDFT10 / REDFT01 kernel.  The emulation backend has none (be_jit_available() = false).
bool be_jit_available();
// be_jit_build returns the number of kernels written to funcs (row: REDFT10, REDFT01 [, 8-bit REDFT10, 8-bit REDFT01 with extras];
// column: REDFT10, REDFT01 [, fused roundtrip with extras]) or a negative error
int be_jit_build(const char *spec_type, int is_col, int extras, const char *incdir, void **funcs, char *log, size_t loglen);
int be_jit_launch(void *func, const void *args, int nwg, int nthr, void *stream);
int be_jit_launch_n(void *func, void **args, int nwg, int nthr, void *stream);      // one pointer per kernel parameter

// small blocks transformed along all their axes in one pass (block_core.h); nwg workgroups of BLOCK_THREADS, `lds` bytes each
bool be_block_supported(int nx, int ny, int nz);
int be_launch_block(const BlockArgs &a, int nwg, size_t lds, void *stream);
// forward -> filter -> inverse of every block in one pass, float or 8-bit samples at either end
int be_launch_block_roundtrip(const BlockRtArgs &a, int nwg, size_t lds, void *stream);

// compile-time-specialised kernels (dct_spec.h / spec_list.h)
struct SpecInfo { int id, nthr, P; size_t lds; int chan = 0; };   // P = C (ROW) or K (COL); chan = G when the interleaved line runs as G channel lines (nthr, lds are theirs)
// DSPFFT_ROW_CHAN=0 keeps such lines in one workgroup (A/B runs)
// (suspended for the duration of a batch that spreads double frames over several streams: another stream's column tiles in the L2s keep
// the channel lines' partial stores from merging -- 0.76 -> 0.82 ms for two 4K frames on two streams)
inline thread_local int g_chan_lines_suspended = 0;
inline bool chan_lines_enabled()
{
	const char *e = getenv("DSPFFT_ROW_CHAN");       // 0: never, 2: also in such batches (A/B runs)
	const int v = e ? atoi(e) : 1;
	return v == 2 || (v != 0 && !g_chan_lines_suspended);
}
// prefer_k > 0 (columns): the entry of that tile width first, where the list has one (slice plans of a clip: spec_list.h)
bool be_find_spec(int is_col, int N, int P, SpecInfo *info, int prefer_k = 0);
int be_launch_spec(int is_col, int id, const PassArgs &a, int nwg, void *stream);
// the same for double samples (spec_list.h DSPFFT_*_SPECS_F64): plain passes only
bool be_find_spec_f64(int is_col, int N, int P, SpecInfo *info);
int be_launch_spec(int is_col, int id, const PassArgsD &a, int nwg, void *stream);
// planar row pass reading u8 (REDFT10) or writing quantised u8 (REDFT01); only specs with C == 1 have it
bool be_spec_has_u8(int row_spec_id);
int be_launch_spec_u8(int row_spec_id, const PassArgs &a, const U8IO &io, int nwg, void *stream);
// out = A(in_a) + B(in_b): two REDFT01 row transforms (same row spec, same output lines) in one launch
int be_launch_row_sum2(int row_spec_id, const PassArgs &a, const PassArgs &b, int nwg, void *stream);
// zoom's x stage on the duo row kernel (dct_duo.h, spec_list.h DSPFFT_ZOOMX_SPECS): lines of M RGB samples from cw coefficients each.
// be_find_zoomx: id or -1.  nsrc = source pixels per slot pair (1: cw <= M/4, 2: cw <= M/2, 4); clip: a.vw < M.
// be_zoomx_tables fills tab ([nsrc][M/2] slots of four floats) for (cw, theta, scale) on the stream -- double-precision arithmetic, one
// entry per thread (zoomx_table_entry)
int be_find_zoomx(int M);
int be_launch_zoomx(int id, const ZoomXArgs &a, int nsrc, bool clip, void *stream);
int be_zoomx_tables(float *tab, int M, int cw, int nsrc, double theta, double scale, void *stream);
// chirp-z rows (dct_czt.h, spec_list.h DSPFFT_CZT_SPECS): be_find_czt returns the id of the smallest listed convolution length P >= need (or -1);
// be_czt_tables fills whichever of the a- (nc), e- (nout) and h- (P entries) tables is non-null; be_launch_czt_spectrum turns the h-table
// (a.atab, a.nc = P, a.in = null) into the slot-ordered spectrum the row kernel multiplies by; be_transpose: out[c][r] = in[r][c]
int be_find_czt(int need, int *P);
int be_launch_czt_rows(int id, const CztArgs &a, void *stream);
int be_launch_czt_spectrum(int id, const CztArgs &a, cf *hspec, void *stream);
int be_czt_tables(cf *atab, cf *etab, cf *htab, int nc, int nout, int P, double omega, double phi, double scale, void *stream);
int be_transpose(float *out, long long out_pitch, const float *in, long long in_pitch, int rows, int cols, void *stream);
// fused column roundtrip: REDFT10 along the tile axis (af), pointwise filter, REDFT01 (ai); both passes share spec `id`
int be_launch_roundtrip(int id, const PassArgs &af, const PassArgs &ai, const MotionFilter &filt, unsigned long long *coded, int nwg, void *stream);

// strided 3-D regions between an 8-bit buffer and a float buffer (motion's block / scaled regions inside a larger embedding,
// motion/motion.c:617-640,756-776), and zeroing (motion.c:619 memset before a block is loaded)
int be_zero(void *p, size_t bytes, void *stream);
int be_region_u8_to_f32(float *dst, const uint8_t *src, const int n[3], const long long sdst[3], const long long ssrc[3], void *stream);
int be_region_f32_to_u8(uint8_t *dst, const float *src, double mul, const int n[3], const long long sdst[3], const long long ssrc[3], void *stream);

// scan orders other than zigzag (scan_core.h, scan_methods.hip).  be_scan_owner_index: step == 0 writes the owner scan index of
// every pixel, step > 0 the frame id index / step with the DC pixel set to 0xFFFFFFFF
int be_scan_owner_index(uint32_t *idx, int method, uint32_t w, uint32_t h, uint64_t step, void *stream);
int be_scan_coords(uint32_t *lin, int method, uint32_t w, uint32_t h, uint64_t first, uint64_t count, uint64_t slots, void *stream);
int be_scan_stamp(uint32_t *ids, const uint32_t *lin, uint64_t n, uint32_t frame, void *stream);
int be_scan_index_to_frame_ids(uint32_t *ids, uint64_t n, uint64_t step, void *stream);
// ranges[2 * (half * ntiles + tile)] = min, [.. + 1] = max owner id of the tile (scan_core.h tile_range_item); halves = 1 or 2
int be_scan_tile_ranges(uint32_t *ranges, const uint32_t *ids, TileRangeGeom g, int halves, void *stream);
int be_scan_tile_eids(void *eids, const uint32_t *ids, TileEidGeom g, void *stream);     // scan_core.h tile_eid_item over every element
int be_download(void *dst, const void *src, size_t bytes, void *stream);                // device -> host, after the stream's work (synchronous)
size_t be_scan_magnitude_work_bytes(uint32_t w, uint32_t h);
int be_scan_magnitude_index(uint32_t *idx, const float *coeffs, uint32_t w, uint32_t h, int ch, double q, void *work, size_t work_bytes, uint32_t *limit, void *stream);

// timing events on a stream (dspfft_execute_many's profiling aid); the emulation backend has none (elapsed = 0)
void *be_event_create();
void be_event_destroy(void *e);
int be_event_record(void *e, void *stream);
int be_event_synchronize(void *e);
int be_event_elapsed_ms(void *a, void *b, float *ms);
// streams of the library's own (dspfft_stream_create): plain non-blocking HIP streams
void *be_stream_create();
void be_stream_destroy(void *s);
int be_stream_synchronize(void *s);
// ordering-only events between streams (dspfft_execute_many_repeat re-joins its streams with them)
void *be_order_event_create();
int be_stream_wait_event(void *stream, void *e);

// outer-radix-2 split of a long column axis (dct_spec.h ColHalfSpec): half-tile column kernels for length N whose inner extent is
// a multiple of the tile width, and the paired row kernel of row spec (N, C); be_find_row_pair returns an id or -1
bool be_find_half_spec(int N, int inner, SpecInfo *info);
int be_find_row_pair(int N, int C);
int be_row_pair_threads(int id);      // the paired kernel may run on another workgroup size than the row spec of the same (N, C)
int be_launch_col_half(int id, const PassArgs &a, int nwg, void *stream);
int be_launch_row_pair(int id, const PassArgs &a, int npairs, void *stream);

// motion's filter over every active element of a buffer of `span` elements (the unfused form of the roundtrip's middle step)
int be_motion_filter(float *buf, const MotionFilter &filt, uint64_t span, unsigned long long *coded, void *stream);

// elementwise helpers (dspfft.h, "device-side helpers")
int be_scan_zigzag(uint32_t *lin, uint32_t w, uint32_t h, uint64_t first, uint64_t count, void *stream);
int be_scan_zigzag_frame_ids(uint32_t *ids, uint32_t w, uint32_t h, uint64_t step, void *stream);
int be_scan_scatter(float *recon, const float *coeffs, const uint32_t *lin, uint64_t count, uint64_t npixels, int channels, void *stream);
int be_accumulate(float *sum, const float *image, uint64_t len, void *stream);
int be_broadcast_dc(float *sum, const float *coeffs, uint64_t npixels, int channels, void *stream);
int be_u8_to_f32(float *dst, const uint8_t *src, uint64_t len, void *stream);
int be_f32_to_u8(uint8_t *dst, const float *src, double mul, uint64_t len, void *stream);

}  // namespace dspfft
