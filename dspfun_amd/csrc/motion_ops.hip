// motion_ops.hip -- the remaining elementwise / selection stages of motion's block loop (motion/motion.c) as device kernels:
//   :617-640  pixel load with the inverse-spectrogram decodes (--ispec shift / flat / copy)
//   :652-668  keep the N coefficients of largest magnitude (--coeff-limit): radix select, no full sort
//   :755-776  output scaling (scalefactor, normalization), spectrogram encodes (--spec abs / shift / flat), clamp + lround
// Scalar math in double (`intermediate` of the reference's motion build, motion/Makefile:1-2).
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <math.h>
#include <stdio.h>

#include "../../include/dspfft.h"
#include "motion_filter.h"

static_assert(dspfft::MOTION_MODE_NONE == DSPFFT_MOTION_NONE && dspfft::MOTION_MODE_ABS == DSPFFT_MOTION_ABS && dspfft::MOTION_MODE_SHIFT == DSPFFT_MOTION_SHIFT &&
              dspfft::MOTION_MODE_FLAT == DSPFFT_MOTION_FLAT && dspfft::MOTION_MODE_COPY == DSPFFT_MOTION_COPY, "motion_filter.h's modes are dspfft.h's");

namespace {

thread_local char g_merr[256] = "";
int mbad(const char *m) { snprintf(g_merr, sizeof g_merr, "%s", m); return -1; }
inline int mgrid(size_t n) { size_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : b > 8192 ? 8192 : b); }

struct Reg { int n[3]; long long mh, mw; };
__device__ inline size_t reg_off(const Reg &r, size_t i) { const size_t x = i % r.n[2], y = (i / r.n[2]) % r.n[1], z = i / ((size_t)r.n[2] * r.n[1]); return (z * r.mh + y) * r.mw + x; }

// PIX = uint8_t (the tool's default) or float (float_pixels: motion.c:623 reads sample * 255, :774 stores pel / 255)
template <class PIX>
__global__ void motion_load_kernel(float *c, const PIX *pix, Reg r, int mode, double ic, double norm)
{
	const size_t total = (size_t)r.n[0] * r.n[1] * r.n[2];
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const size_t o = reg_off(r, i);
		double pel;
		if constexpr (sizeof(PIX) == 1) pel = (double)pix[o]; else pel = (double)(pix[o] * 255.0f) ;   // :623 float * int: a float product
		c[o] = (float)dspfft::motion_load_pel(pel, mode, ic, norm);                                     // :627-637
	}
}

template <class PIX>
__global__ void motion_store_kernel(PIX *pix, const float *c, Reg r, int mode, double scalefactor, double norm, double cc)
{
	const size_t total = (size_t)r.n[0] * r.n[1] * r.n[2];
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const size_t o = reg_off(r, i);
		const double pel = dspfft::motion_store_pel((double)c[o], mode, scalefactor, norm, cc);         // :759-771
		if constexpr (sizeof(PIX) == 1) pix[o] = pel > 255 ? 255 : pel < 0 ? 0 : (uint8_t)lround(pel);  // :776
		else pix[o] = (float)(pel / 255);                                                               // :774
	}
}

// ---- top-N by magnitude: radix select on the bits of |c| (monotone for non-negative floats) ----
struct SelState { uint32_t prefix, mask, remaining, pad; };       // keys matching (key & mask) == prefix are still candidates
__global__ void topn_hist_kernel(uint32_t *hist, const float *c, size_t n, const SelState *st, int shift)
{
	__shared__ uint32_t h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t prefix = st->prefix, mask = st->mask;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const uint32_t k = __float_as_uint(fabsf(c[i]));
		if ((k & mask) == prefix) atomicAdd(&h[(k >> shift) & 255], 1u);
	}
	__syncthreads();
	if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// one thread: walk the 256 bins from the top, find the bin holding the `remaining`-th largest candidate
__global__ void topn_pick_kernel(uint32_t *hist, SelState *st, int shift)
{
	if (threadIdx.x || blockIdx.x) return;
	uint32_t rem = st->remaining;
	int b = 255;
	for (; b > 0; b--) { if (hist[b] >= rem) break; rem -= hist[b]; }
	st->prefix |= (uint32_t)b << shift;
	st->mask |= 255u << shift;
	st->remaining = rem;                      // how many of the candidates in bin b (and, after the last pass, equal to the threshold) to keep
	for (int i = 0; i < 256; i++) hist[i] = 0;
}
__global__ void topn_flag_kernel(uint32_t *tie, const float *c, size_t n, const SelState *st)
{
	const uint32_t T = st->prefix;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		tie[i] = __float_as_uint(fabsf(c[i])) == T ? 1u : 0u;
}
__global__ void topn_apply_kernel(float *c, const uint32_t *rank, size_t n, const SelState *st)
{
	const uint32_t T = st->prefix, keep_ties = st->remaining;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const uint32_t k = __float_as_uint(fabsf(c[i]));
		if (!(k > T || (k == T && rank[i] < keep_ties))) c[i] = 0.f;
	}
}

size_t scan_temp(size_t n)
{
	size_t b = 0;
	uint32_t *p = nullptr;
	(void)rocprim::exclusive_scan(nullptr, b, p, p, 0u, n, rocprim::plus<uint32_t>(), (hipStream_t)nullptr);
	return b + 256;
}

}  // namespace

extern "C" const char *dspfft_motion_last_error(void) { return g_merr; }

extern "C" int dspfft_motion_load_u8(float *d_coeffs, const uint8_t *d_pix, const int n[3], const int minbuf_hw[2], int ispec_mode, double ic, double normalization, void *stream)
{
	if (!d_coeffs || !d_pix || !n || !minbuf_hw || n[0] < 1 || n[1] < 1 || n[2] < 1 || minbuf_hw[0] < n[1] || minbuf_hw[1] < n[2]) return mbad("bad arguments");
	if (ispec_mode != DSPFFT_MOTION_NONE && ispec_mode != DSPFFT_MOTION_SHIFT && ispec_mode != DSPFFT_MOTION_FLAT && ispec_mode != DSPFFT_MOTION_COPY) return mbad("ispec mode: none, shift, flat or copy");
	Reg r; r.n[0] = n[0]; r.n[1] = n[1]; r.n[2] = n[2]; r.mh = minbuf_hw[0]; r.mw = minbuf_hw[1];
	hipLaunchKernelGGL(motion_load_kernel<uint8_t>, dim3(mgrid((size_t)n[0] * n[1] * n[2])), dim3(256), 0, (hipStream_t)stream, d_coeffs, d_pix, r, ispec_mode, ic, normalization);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int dspfft_motion_load_f32(float *d_coeffs, const float *d_pix, const int n[3], const int minbuf_hw[2], int ispec_mode, double ic, double normalization, void *stream)
{
	if (!d_coeffs || !d_pix || !n || !minbuf_hw || n[0] < 1 || n[1] < 1 || n[2] < 1 || minbuf_hw[0] < n[1] || minbuf_hw[1] < n[2]) return mbad("bad arguments");
	if (ispec_mode != DSPFFT_MOTION_NONE && ispec_mode != DSPFFT_MOTION_SHIFT && ispec_mode != DSPFFT_MOTION_FLAT && ispec_mode != DSPFFT_MOTION_COPY) return mbad("ispec mode: none, shift, flat or copy");
	Reg r; r.n[0] = n[0]; r.n[1] = n[1]; r.n[2] = n[2]; r.mh = minbuf_hw[0]; r.mw = minbuf_hw[1];
	hipLaunchKernelGGL(motion_load_kernel<float>, dim3(mgrid((size_t)n[0] * n[1] * n[2])), dim3(256), 0, (hipStream_t)stream, d_coeffs, d_pix, r, ispec_mode, ic, normalization);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int dspfft_motion_store_u8(uint8_t *d_pix, const float *d_coeffs, const int n[3], const int minbuf_hw[2], int spec_mode,
                                      double scalefactor, double normalization, double c, void *stream)
{
	if (!d_coeffs || !d_pix || !n || !minbuf_hw || n[0] < 1 || n[1] < 1 || n[2] < 1 || minbuf_hw[0] < n[1] || minbuf_hw[1] < n[2]) return mbad("bad arguments");
	if (spec_mode < DSPFFT_MOTION_NONE || spec_mode > DSPFFT_MOTION_COPY) return mbad("spec mode: none, abs, shift, flat or copy");
	Reg r; r.n[0] = n[0]; r.n[1] = n[1]; r.n[2] = n[2]; r.mh = minbuf_hw[0]; r.mw = minbuf_hw[1];
	hipLaunchKernelGGL(motion_store_kernel<uint8_t>, dim3(mgrid((size_t)n[0] * n[1] * n[2])), dim3(256), 0, (hipStream_t)stream, d_pix, d_coeffs, r, spec_mode, scalefactor, normalization, c);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int dspfft_motion_store_f32(float *d_pix, const float *d_coeffs, const int n[3], const int minbuf_hw[2], int spec_mode,
                                       double scalefactor, double normalization, double c, void *stream)
{
	if (!d_coeffs || !d_pix || !n || !minbuf_hw || n[0] < 1 || n[1] < 1 || n[2] < 1 || minbuf_hw[0] < n[1] || minbuf_hw[1] < n[2]) return mbad("bad arguments");
	if (spec_mode < DSPFFT_MOTION_NONE || spec_mode > DSPFFT_MOTION_COPY) return mbad("spec mode: none, abs, shift, flat or copy");
	Reg r; r.n[0] = n[0]; r.n[1] = n[1]; r.n[2] = n[2]; r.mh = minbuf_hw[0]; r.mw = minbuf_hw[1];
	hipLaunchKernelGGL(motion_store_kernel<float>, dim3(mgrid((size_t)n[0] * n[1] * n[2])), dim3(256), 0, (hipStream_t)stream, d_pix, d_coeffs, r, spec_mode, scalefactor, normalization, c);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" size_t dspfft_motion_topn_work_bytes(size_t count) { return 2 * ((count * 4 + 255) & ~(size_t)255) + 4096 + scan_temp(count); }

extern "C" int dspfft_motion_topn(float *d_coeffs, size_t count, size_t keep, void *d_work, size_t work_bytes, void *stream)
{
	if (!d_coeffs || !d_work || !count) return mbad("bad arguments");
	if (work_bytes < dspfft_motion_topn_work_bytes(count)) return mbad("work buffer too small: dspfft_motion_topn_work_bytes");
	if (count >= (1ull << 32)) return mbad("top-N select addresses the buffer with 32-bit counts");
	hipStream_t s = (hipStream_t)stream;
	if (keep >= count) return 0;
	if (!keep) return hipMemsetAsync(d_coeffs, 0, count * 4, s) == hipSuccess ? 0 : -4;
	const size_t slab = (count * 4 + 255) & ~(size_t)255;
	char *base = (char *)d_work;
	SelState *st = (SelState *)base;
	uint32_t *hist = (uint32_t *)(base + 1024), *tie = (uint32_t *)(base + 4096), *rank = (uint32_t *)(base + 4096 + slab);
	void *temp = base + 4096 + 2 * slab;
	size_t tb = scan_temp(count);
	SelState init = {0u, 0u, (uint32_t)keep, 0u};
	if (hipMemcpyAsync(st, &init, sizeof init, hipMemcpyHostToDevice, s) != hipSuccess || hipMemsetAsync(hist, 0, 1024, s) != hipSuccess) return -4;
	for (int shift = 24; shift >= 0; shift -= 8) {
		hipLaunchKernelGGL(topn_hist_kernel, dim3(mgrid(count)), dim3(256), 0, s, hist, d_coeffs, count, st, shift);
		hipLaunchKernelGGL(topn_pick_kernel, dim3(1), dim3(64), 0, s, hist, st, shift);
	}
	// elements equal to the threshold: the first `remaining` of them in buffer order are kept (the reference leaves ties to qsort)
	hipLaunchKernelGGL(topn_flag_kernel, dim3(mgrid(count)), dim3(256), 0, s, tie, d_coeffs, count, st);
	if (rocprim::exclusive_scan(temp, tb, tie, rank, 0u, count, rocprim::plus<uint32_t>(), s) != hipSuccess) return -4;
	hipLaunchKernelGGL(topn_apply_kernel, dim3(mgrid(count)), dim3(256), 0, s, d_coeffs, rank, count, st);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}
