// pointwise.hip -- the elementwise stages the tools run either side of the transform,
// as device kernels so buffers never leave HBM between the DCT and its display encoding / filtering:
//   spec/spec.c:81-139   coefficient -> spectrogram encoding (gain, range, log/linear scale, sign mapping)
//   spec/ispec.c:84-151  the inverse decoding (without the separate sign-map image of :91-99)
//   motion/motion.c:683-744  band-pass damp / boost, threshold, DC preservation, quantisation
// Scalar math follows the reference's `intermediate` = double.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

#include "../../include/dspfft.h"
#include "motion_filter.h"

namespace {

enum { RANGE_ONE = 0, RANGE_DC = 1, RANGE_DCS = 2 };
enum { SCALE_LOG = 0, SCALE_LINEAR = 1 };
enum { SIGN_ABS = 0, SIGN_SHIFT = 1, SIGN_SATURATE = 2, SIGN_RETAIN = 3 };

// spec.c:92-110 (+ :115 log1p): per-channel divisor from the (gain-multiplied) DC terms of the FIRST pixel
__device__ inline void spec_divisors(double *mx, const float *f, int d, double gain, int rangetype, int scaletype)
{
	float m[8];
	if (rangetype == RANGE_ONE) for (int z = 0; z < d; z++) m[z] = (float)gain;
	else if (rangetype == RANGE_DC) {
		float best = (float)(f[0] * gain);
		for (int z = 1; z < d; z++) { float v = (float)(f[z] * gain); if (v > best) best = v; }
		for (int z = 0; z < d; z++) m[z] = best;
	} else for (int z = 0; z < d; z++) m[z] = (float)(f[z] * gain);
	for (int z = 0; z < d; z++) mx[z] = scaletype == SCALE_LOG ? (double)log1pf(m[z]) : (double)m[z];   // mc(log1p) on coeff max[] (spec.c:115)
}
__device__ inline float spec_encode_one(float fi, size_t i, int d, double gain, const double *mx, int scaletype, int signtype)
{
	float v = (float)(fi * gain);                                                       // spec.c:88-89
	const float m = (float)mx[i % d];
	if (scaletype == SCALE_LOG) v = (float)(copysign(log1p((double)fabsf(v)), (double)v) / m);   // :117
	else v = v / m;                                                                     // :121
	if (signtype == SIGN_ABS) v = fabsf(v);                                             // :128
	else if (signtype == SIGN_SHIFT) v = (float)(((double)v / 2. + 0.5) * 254 / 255);   // :132
	else if (signtype == SIGN_SATURATE) { if (i >= (size_t)d) v = !signbit(v); }        // :135-136
	return v;
}
// FIRST = 0: every sample but the first pixel's (which this launch only reads, so every thread can derive the divisors from
// it -- no device scratch, no allocation: the call can sit in a captured graph); FIRST = 1: one block, the first pixel itself
template <int FIRST>
__global__ void spec_encode_kernel(float *f, size_t len, int d, double gain, int rangetype, int scaletype, int signtype)
{
	double mx[8];
	spec_divisors(mx, f, d, gain, rangetype, scaletype);
	if (FIRST) {
		__syncthreads();                                    // all threads of the single block have read the DC terms
		if ((int)threadIdx.x < d) f[threadIdx.x] = spec_encode_one(f[threadIdx.x], threadIdx.x, d, gain, mx, scaletype, signtype);
		return;
	}
	for (size_t i = (size_t)d + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
		f[i] = spec_encode_one(f[i], i, d, gain, mx, scaletype, signtype);
}

__global__ void ispec_decode_kernel(float *f, size_t len, int d, double gain, double m0, double m1, double m2, double m3,
                                    int scaletype, int signtype, int restore_dc, double dc0, double dc1, double dc2, double dc3)
{
	const double mx[4] = {m0, m1, m2, m3}, dc[4] = {dc0, dc1, dc2, dc3};
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) {
		float v = f[i];
		if (signtype == SIGN_SHIFT) v = (float)(((double)v * 255. / 254 - 0.5) * 2);        // ispec.c:102
		else if (signtype == SIGN_SATURATE) { if (i >= (size_t)d) v = v * 2 - 1; }          // :106
		const double m = mx[i % d];
		if (scaletype == SCALE_LOG) v = (float)copysign(expm1((double)fabsf((float)(v * m))), (double)v);   // :142
		else v = (float)(v * m);                                                             // :146
		v = (float)(v / gain);                                                               // :150-151
		if (restore_dc && i < (size_t)d) v = (float)dc[i];                                   // :161-163 (the fused plan scaling is 1 at the corner)
		f[i] = v;
	}
}

// spec/ispec.c:91-99: signs of an `abs` spectrogram from its companion sign-map image (8 bits per sample, what spec's
// `saturate` sign type writes): f[i] = copysign(f[i], map[i] - 128) for every sample but the first pixel's
__global__ void ispec_signmap_kernel(float *f, const uint8_t *map, size_t len, int d)
{
	for (size_t i = (size_t)d + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x)
		f[i] = copysignf(f[i], (float)((int)map[i] - 128));
}

__global__ void motion_filter_kernel(float *c, dspfft::MotionFilter p, unsigned long long *coded)
{
	const size_t total = (size_t)p.ad * p.ah * p.aw;
	unsigned long long mine = 0;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int x = (int)(i % p.aw), y = (int)((i / p.aw) % p.ah), z = (int)(i / ((size_t)p.aw * p.ah));
		const size_t o = ((size_t)z * p.mh + y) * p.mw + x;
		c[o] = dspfft::motion_filter_at(p, z, y, x, c[o], mine);
	}
	if (coded) {      // one atomic per wave
		for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
		if ((threadIdx.x & 63) == 0 && mine) atomicAdd(coded, mine);
	}
}

// scan/scan.c:20-28 basis matrix entries B_N[k][j] = j ? 2 cos(pi j (k + 1/2) / N) : 1, for the selected j only
__global__ void pruned_basis_kernel(float *by, float *bx, const uint32_t *lin, int ncoords, int w, int h)
{
	const int total = ncoords * (w + h);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
		const int n = i / (w + h), r = i - n * (w + h);
		const uint32_t p = lin[n];
		const int cy = (int)(p / (uint32_t)w), cx = (int)(p % (uint32_t)w);
		if (r < h) by[n * h + r] = cy ? (float)(2.0 * cos(M_PI * cy * (r + 0.5) / h)) : 1.f;
		else { const int x = r - h; bx[n * w + x] = cx ? (float)(2.0 * cos(M_PI * cx * (x + 0.5) / w)) : 1.f; }
	}
}
// scan/scan.c:30-41 + :451-459: sum += sum_n coeff[n] * B_h[y][cy_n] * B_w[x][cx_n]   (DC pixel excluded by the caller's list)
__global__ void pruned_accumulate_kernel(float *sum, const float *coeffs, const uint32_t *lin, const float *by, const float *bx,
                                         int ncoords, int w, int h, int ch)
{
	const size_t total = (size_t)w * h * ch;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int z = (int)(i % ch);
		const size_t p = i / ch;
		const int x = (int)(p % w), y = (int)(p / w);
		float acc = 0.f;
		for (int n = 0; n < ncoords; n++) {
			const float c = coeffs[(size_t)lin[n] * ch + z] * by[n * h + y] * bx[n * w + x];
			acc = n ? acc + c : c;
		}
		sum[i] += acc;
	}
}

thread_local char g_perr[256] = "";
int bad(const char *m) { snprintf(g_perr, sizeof g_perr, "%s", m); return -1; }
inline int grid_for(size_t n) { size_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : b > 4096 ? 4096 : b); }

}  // namespace

extern "C" const char *dspfft_pointwise_last_error(void) { return g_perr; }

extern "C" int dspfft_spec_encode(float *d_f, size_t npixels, int channels, double gain, int rangetype, int scaletype, int signtype, void *stream)
{
	if (!d_f || channels < 1 || channels > 8 || rangetype < 0 || rangetype > 2 || scaletype < 0 || scaletype > 1 || signtype < 0 || signtype > 3) return bad("bad arguments");
	hipStream_t s = (hipStream_t)stream;
	const size_t len = npixels * channels;
	if (len > (size_t)channels)
		hipLaunchKernelGGL(spec_encode_kernel<0>, dim3(grid_for(len - channels)), dim3(256), 0, s, d_f, len, channels, gain, rangetype, scaletype, signtype);
	hipLaunchKernelGGL(spec_encode_kernel<1>, dim3(1), dim3(64), 0, s, d_f, len, channels, gain, rangetype, scaletype, signtype);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int dspfft_ispec_decode(float *d_f, size_t npixels, int channels, double gain, int rangetype, int scaletype, int signtype,
                                   const double *dc, int restore_dc, void *stream)
{
	if (!d_f || channels < 1 || channels > 4 || rangetype < 0 || rangetype > 2 || scaletype < 0 || scaletype > 1 || signtype < 0 || signtype > 3) return bad("bad arguments (up to 4 channels)");
	if ((rangetype != RANGE_ONE || restore_dc) && !dc) return bad("the DC values from the spectrogram header are required for range dc/dcs and for DC restoration");
	double mx[4] = {0, 0, 0, 0}, d4[4] = {0, 0, 0, 0};
	for (int z = 0; z < channels && dc; z++) d4[z] = dc[z];
	// ispec.c:118-134 (max[] is `coeff`, i.e. float) and :138-139 (log1p in double)
	float m[4];
	if (rangetype == RANGE_ONE) for (int z = 0; z < channels; z++) m[z] = (float)gain;
	else if (rangetype == RANGE_DC) {
		float best = (float)(dc[0] * gain);
		for (int z = 1; z < channels; z++) if (dc[z] * gain > best) best = (float)(dc[z] * gain);
		for (int z = 0; z < channels; z++) m[z] = best;
	} else for (int z = 0; z < channels; z++) m[z] = (float)(dc[z] * gain);
	for (int z = 0; z < channels; z++) mx[z] = scaletype == SCALE_LOG ? (double)(float)log1p((double)m[z]) : (double)m[z];
	const size_t len = npixels * channels;
	hipLaunchKernelGGL(ispec_decode_kernel, dim3(grid_for(len)), dim3(256), 0, (hipStream_t)stream, d_f, len, channels, gain,
	                   mx[0], mx[1], mx[2], mx[3], scaletype, signtype, restore_dc, d4[0], d4[1], d4[2], d4[3]);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int dspfft_ispec_signmap(float *d_f, const uint8_t *d_map, size_t npixels, int channels, void *stream)
{
	if (!d_f || !d_map || channels < 1) return bad("bad arguments");
	const size_t len = npixels * channels;
	if (len <= (size_t)channels) return 0;
	hipLaunchKernelGGL(ispec_signmap_kernel, dim3(grid_for(len - channels)), dim3(256), 0, (hipStream_t)stream, d_f, d_map, len, channels);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int dspfft_motion_filter(float *d_coeffs, const int active[3], const int minbuf_hw[2], const int band_begin[3], const int band_end[3],
                                    float damp, float boost, float threshold_lo, float threshold_hi, int preserve_dc, float grey_add,
                                    float quantizer, unsigned long long *d_coeffs_coded, void *stream)
{
	if (!d_coeffs || !active || !minbuf_hw || !band_begin || !band_end || preserve_dc < 0 || preserve_dc > 2) return bad("bad arguments");
	dspfft::MotionFilter p;
	p.ad = active[0]; p.ah = active[1]; p.aw = active[2]; p.mh = minbuf_hw[0]; p.mw = minbuf_hw[1];
	p.b0d = band_begin[0]; p.b0h = band_begin[1]; p.b0w = band_begin[2]; p.b1d = band_end[0]; p.b1h = band_end[1]; p.b1w = band_end[2];
	p.damp = damp; p.boost = boost; p.thr_lo = threshold_lo; p.thr_hi = threshold_hi; p.preserve_dc = preserve_dc; p.grey_add = grey_add; p.quantizer = quantizer;
	p.enabled = 1; dspfft::motion_filter_set_divs(p, p.ad > 0 ? p.ad : 1);
	const size_t total = (size_t)p.ad * p.ah * p.aw;
	if (!total) return 0;
	hipLaunchKernelGGL(motion_filter_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, d_coeffs, p, d_coeffs_coded);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

/* scan's pruned inverse (scan/scan.c:20-41,449): for frames that add only a few coefficients, the direct sum of
 * rank-1 basis products replaces the two-pass REDFT01 -- one read + one write of `sum`, no intermediate. */
extern "C" size_t dspfft_scan_pruned_work_floats(int ncoords, int w, int h) { return (size_t)(ncoords > 0 ? ncoords : 0) * ((size_t)w + h); }

/* the same with the basis table in a caller-provided buffer: no allocation, so a scan loop on the pruned path can be captured */
extern "C" int dspfft_scan_pruned_accumulate_ws(float *d_sum, const float *d_coeffs, const uint32_t *d_lin, int ncoords,
                                                int w, int h, int channels, float *d_work, void *stream)
{
	if (!d_sum || !d_coeffs || (!d_lin && ncoords) || ncoords < 0 || w < 1 || h < 1 || channels < 1 || (!d_work && ncoords)) return bad("bad arguments");
	if (!ncoords) return 0;
	hipStream_t s = (hipStream_t)stream;
	float *by = d_work, *bx = d_work + (size_t)ncoords * h;
	hipLaunchKernelGGL(pruned_basis_kernel, dim3(grid_for((size_t)ncoords * (w + h))), dim3(256), 0, s, by, bx, d_lin, ncoords, w, h);
	hipLaunchKernelGGL(pruned_accumulate_kernel, dim3(grid_for((size_t)w * h * channels)), dim3(256), 0, s, d_sum, d_coeffs, d_lin, by, bx, ncoords, w, h, channels);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int dspfft_scan_pruned_accumulate(float *d_sum, const float *d_coeffs, const uint32_t *d_lin, int ncoords,
                                             int w, int h, int channels, void *stream)
{
	if (!d_sum || !d_coeffs || (!d_lin && ncoords) || ncoords < 0 || w < 1 || h < 1 || channels < 1) return bad("bad arguments");
	if (!ncoords) return 0;
	hipStream_t s = (hipStream_t)stream;
	float *tab = nullptr;
	if (hipMallocAsync((void **)&tab, sizeof(float) * (size_t)ncoords * (w + h), s) != hipSuccess) return bad("hipMallocAsync failed");
	float *by = tab, *bx = tab + (size_t)ncoords * h;
	hipLaunchKernelGGL(pruned_basis_kernel, dim3(grid_for((size_t)ncoords * (w + h))), dim3(256), 0, s, by, bx, d_lin, ncoords, w, h);
	hipLaunchKernelGGL(pruned_accumulate_kernel, dim3(grid_for((size_t)w * h * channels)), dim3(256), 0, s, d_sum, d_coeffs, d_lin, by, bx, ncoords, w, h, channels);
	(void)hipFreeAsync(tab, s);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}
