/*
 * scan_dev -- scan's main loop (scan/scan.c:289-298 forward + normalisation, :377-383 DC broadcast, :421-459 per-frame
 * scatter / inverse / accumulate) with EVERY buffer resident on the GPU and every scan method generated there
 * (include/dspfft.h "the other scan methods on the device"): one fused masked-accumulate execution per output frame, no
 * PCIe traffic inside the loop.  The host-pointer drop-in of the same loop is host/scan_gpu.c.
 *
 *   scan_dev in.{ppm,pf} out.pf [step] [method]
 *     method: a prefix of horizontal vertical zigzag row column diagonal mirror box ibox radial iradial (scan_methods.c:581-591),
 *             magnitude[:qfactor] (scan_methods.c:240-296), file:<path> (scan_methods.c:393-410, either serialisation), or
 *             random[:seed] (scan_methods.c:210-228: the permutation is drawn on the host with libc rand(), as the tool draws it).
 *             evalxy / evali (scan_methods.c:186-201,333-391) need libavutil's expression evaluator and are not mirrored: write the order
 *             they would give to a file and pass file:<path>.
 * Output: the final `sum` image; on stderr the number of frames and max|sum - input| (0 up to rounding when the method visits
 * every pixel exactly once).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <dspfft.h>
#include <hip/hip_runtime_api.h>
#include "precision.h"
#include "rawio.h"
#include "scan_orders.h"

#define HIP(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 1; } } while (0)
#define DSP(x) do { if (x) { fprintf(stderr, "dspfft: %s (%s:%d)\n", dspfft_last_error(), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char *argv[])
{
	if (argc < 3) { fprintf(stderr, "usage: %s <in> <out.pf> [step] [method]\n", argv[0]); return 2; }
	size_t width, height;
	const int channels = 3;
	float *pix;
	if (read_image(argv[1], &width, &height, &pix)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
	const char *mname = argc > 4 ? argv[4] : "zigzag";
	const uint32_t w = (uint32_t)width, h = (uint32_t)height;
	const size_t npix = width * height, n = npix * channels;

	float *d_coeffs, *d_sum, *d_work;
	uint32_t *d_ids;
	HIP(hipMalloc((void **)&d_coeffs, n * 4)); HIP(hipMalloc((void **)&d_sum, n * 4)); HIP(hipMalloc((void **)&d_work, n * 4));
	HIP(hipMalloc((void **)&d_ids, npix * 4));
	HIP(hipMemcpy(d_coeffs, pix, n * 4, hipMemcpyHostToDevice));

	dspfft_plan fwd, inv;
	const int dims[2] = {(int)height, (int)width}, k10[2] = {DSPFFT_REDFT10, DSPFFT_REDFT10}, k01[2] = {DSPFFT_REDFT01, DSPFFT_REDFT01};
	DSP(dspfft_plan_many_r2r(&fwd, 2, dims, channels, NULL, channels, 1, NULL, channels, 1, k10));      /* scan.c:292 */
	DSP(dspfft_plan_set_scale(fwd, 1.0f / (4.0f * width * height)));                                   /* scan.c:296-298 fused */
	DSP(dspfft_plan_many_r2r(&inv, 2, dims, channels, NULL, channels, 1, NULL, channels, 1, k01));      /* scan.c:359 */
	DSP(dspfft_execute(fwd, d_coeffs, d_coeffs, NULL));

	/* ---- the scan order -> what the frame loop needs ---- */
	uint64_t limit = 0, slots = 0;
	int method = -1, per_frame_lists = 0;
	uint32_t *d_lin = NULL;                 /* per-frame coordinate lists (box, or a file whose indices share pixels) */
	struct scan_order_list fl;
	memset(&fl, 0, sizeof fl);
	if (!strncmp(mname, "magnitude", 9)) {
		const double q = mname[9] == ':' ? strtod(mname + 10, NULL) : 0.0;
		void *d_mw;
		const size_t wb = dspfft_scan_magnitude_work_bytes(w, h);
		uint32_t lim32;
		HIP(hipMalloc(&d_mw, wb));
		DSP(dspfft_scan_magnitude_index(d_ids, d_coeffs, w, h, channels, q, d_mw, wb, &lim32, NULL));
		HIP(hipFree(d_mw));
		limit = lim32;
	} else if (!strncmp(mname, "file:", 5) || !strncmp(mname, "random", 6)) {
		if (mname[0] == 'r') {
			const unsigned int seed = mname[6] == ':' ? (unsigned int)strtoul(mname + 7, NULL, 10) : (unsigned int)time(NULL);   /* scan_methods.c:216 */
			if (scan_order_random(width, height, seed, &fl)) { fprintf(stderr, "cannot draw the random scan order\n"); return 1; }
		} else {
			FILE *f = fopen(mname + 5, "r");
			if (!f || scan_order_read_file(f, width, height, &fl)) { fprintf(stderr, "cannot read the scan order %s\n", mname + 5); return 1; }
			fclose(f);
		}
		limit = fl.limit; slots = fl.max_interval;
		/* a pixel listed under several indices needs per-frame lists; otherwise one owner-index array does */
		uint32_t *owner = malloc(npix * 4);
		memset(owner, 0xff, npix * 4);
		for (size_t i = 0; i < fl.limit && !per_frame_lists; i++)
			for (size_t k = fl.offset[i]; k < fl.offset[i + 1]; k++) {
				const size_t p = fl.yx[k][0] * width + fl.yx[k][1];
				if (owner[p] != 0xffffffffu && owner[p] != i) { per_frame_lists = 1; break; }
				owner[p] = (uint32_t)i;
			}
		if (!per_frame_lists) HIP(hipMemcpy(d_ids, owner, npix * 4, hipMemcpyHostToDevice));    /* unlisted pixels keep 0xFFFFFFFF: never reconstructed */
		free(owner);
	} else {
		method = scan_order_find_prefix(mname);                                               /* scan.c:176 scan_method_find_prefix */
		if (method < 0) { fprintf(stderr, "unknown scan method %s\n", mname); return 2; }
		limit = dspfft_scan_limit(method, w, h);
		slots = dspfft_scan_coord_slots(method, w, h);
		per_frame_lists = method == DSPFFT_SCAN_BOX;
	}
	size_t step = argc > 3 ? strtoul(argv[3], NULL, 10) : (limit + 31) / 32;
	if (!step) step = 1;
	const size_t nframes = (limit + step - 1) / step;                                            /* scan.c:347-348 */
	if (per_frame_lists) {
		HIP(hipMalloc((void **)&d_lin, (size_t)step * (slots ? slots : 1) * 4));
		HIP(hipMemset(d_ids, 0xff, npix * 4));
	} else if (method >= 0) DSP(dspfft_scan_frame_ids(d_ids, method, w, h, step, NULL));
	else DSP(dspfft_scan_index_to_frame_ids(d_ids, npix, step, NULL));                            /* magnitude / file: index -> frame */
	/* owner ids that stay put over the frames: let the fused step skip the column tiles a frame does not touch (box restamps its ids) */
	if (!per_frame_lists) DSP(dspfft_plan_scan_prepare(inv, d_ids, channels, NULL));

	DSP(dspfft_broadcast_dc(d_sum, d_coeffs, npix, channels, NULL));                              /* scan.c:377-383 */
	uint32_t *h_lin = per_frame_lists && method < 0 ? malloc((size_t)step * (slots ? slots : 1) * 4) : NULL;
	for (size_t i = 0; i < nframes; i++) {                                                        /* scan.c:421-459 */
		const size_t lo = i * step, hi = lo + step < limit ? lo + step : limit;
		if (per_frame_lists) {
			size_t cnt;
			if (method >= 0) { DSP(dspfft_scan_coords(d_lin, method, w, h, lo, hi - lo, NULL)); cnt = (hi - lo) * slots; }
			else {
				cnt = fl.offset[hi] - fl.offset[lo];
				for (size_t k = 0; k < cnt; k++) h_lin[k] = (uint32_t)(fl.yx[fl.offset[lo] + k][0] * width + fl.yx[fl.offset[lo] + k][1]);
				HIP(hipMemcpy(d_lin, h_lin, cnt * 4, hipMemcpyHostToDevice));
			}
			DSP(dspfft_scan_stamp(d_ids, d_lin, cnt, (uint32_t)i, NULL));
		}
		DSP(dspfft_execute_masked_accumulate(inv, d_coeffs, d_work, d_sum, d_ids, (uint32_t)i, channels, NULL));
	}
	float *sum = malloc(n * 4);
	HIP(hipMemcpy(sum, d_sum, n * 4, hipMemcpyDeviceToHost));
	double err = 0;
	for (size_t j = 0; j < n; j++) { const double e = fabs((double)sum[j] - pix[j]); if (e > err) err = e; }
	fprintf(stderr, "method %s: %zu scan indices, %zu frames of %zu, device-resident; max|sum-input| = %.3e\n", mname, (size_t)limit, nframes, step, err);
	const int rc = write_pf(argv[2], width, height, sum);
	dspfft_destroy_plan(fwd); dspfft_destroy_plan(inv);
	hipFree(d_coeffs); hipFree(d_sum); hipFree(d_work); hipFree(d_ids); hipFree(d_lin);
	free(sum); free(pix); free(h_lin); scan_order_list_free(&fl);
	return rc;
}
