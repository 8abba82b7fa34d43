// scan_methods.hip -- device-side generators for the scan orders other than zigzag (scan_core.h) and the helpers that turn
// them into per-frame masks for the fused scan step (dspfft_execute_masked_accumulate), so that scan's frame loop
// (scan/scan.c:421-459) stays on the GPU for every method:
//   single-owner methods  -> owner index per pixel -> frame id per pixel (index / step), computed once
//   box (pixels shared by several scan indices, out-of-range first leg) -> per frame: coordinate list -> stamp into the id array
//   magnitude (scan_methods.c:240-296) -> sort keys on the device, stable radix sort (rocPRIM), run-change prefix sum
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "backend.h"
#include "scan_core.h"

namespace dspfft {

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
static inline int sgrid(uint64_t n) { uint64_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

__global__ void owner_index_kernel(uint32_t *idx, int method, uint32_t w, uint32_t h, uint64_t step)
{
	const uint64_t n = (uint64_t)w * h;
	for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t y = p / w, x = p - y * w;
		const uint64_t i = scan_owner_index(method, w, h, y, x);
		idx[p] = step ? (p ? (uint32_t)(i / step) : SCAN_NONE) : (uint32_t)i;      // frame ids: the DC pixel never matches (scan.c:445)
	}
}

__global__ void coords_kernel(uint32_t *lin, int method, uint32_t w, uint32_t h, uint64_t first, uint64_t count, uint64_t slots)
{
	const uint64_t n = count * slots;
	for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t k = t / slots, j = t - k * slots, i = first + k;
		lin[t] = j < scan_interval(method, w, h, i) ? scan_coord_lin(method, w, h, i, j) : SCAN_NONE;
	}
}

__global__ void stamp_kernel(uint32_t *ids, const uint32_t *lin, uint64_t n, uint32_t frame)
{
	for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t p = lin[t];
		if (p != SCAN_NONE && p != 0) ids[p] = frame;               // the DC pixel is pre-added and always cleared (scan.c:377-383,445)
	}
}

__global__ void index_to_frame_kernel(uint32_t *ids, uint64_t n, uint64_t step)
{
	for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x)
		ids[p] = p ? (uint32_t)(ids[p] / step) : SCAN_NONE;
}

__global__ void magnitude_keys_kernel(float *keys, uint32_t *vals, const float *coeffs, uint32_t w, uint32_t h, int ch, double q)
{
	const uint64_t n = (uint64_t)w * h;
	for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t y = p / w, x = p - y * w;
		keys[p] = scan_magnitude_key(coeffs + p * ch, ch, y, x, q);
		vals[p] = (uint32_t)p;
	}
}
// a[k] = 1 where the sorted key changes (a[0] = 1: the reference starts from last_val = -1), scan_methods.c:268-276
__global__ void magnitude_flags_kernel(uint32_t *flags, const float *keys, uint64_t n)
{
	for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x)
		flags[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1u : 0u;
}
__global__ void magnitude_scatter_kernel(uint32_t *idx, const uint32_t *sorted, const uint32_t *j, uint64_t n)
{
	for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) idx[sorted[k]] = j[k];
}

int be_scan_owner_index(uint32_t *idx, int method, uint32_t w, uint32_t h, uint64_t step, void *stream)
{
	hipLaunchKernelGGL(owner_index_kernel, dim3(sgrid((uint64_t)w * h)), dim3(256), 0, (hipStream_t)stream, idx, method, w, h, step);
	HIPCHK(hipGetLastError());
	return 0;
}
int be_scan_coords(uint32_t *lin, int method, uint32_t w, uint32_t h, uint64_t first, uint64_t count, uint64_t slots, void *stream)
{
	if (!count || !slots) return 0;
	hipLaunchKernelGGL(coords_kernel, dim3(sgrid(count * slots)), dim3(256), 0, (hipStream_t)stream, lin, method, w, h, first, count, slots);
	HIPCHK(hipGetLastError());
	return 0;
}
int be_scan_stamp(uint32_t *ids, const uint32_t *lin, uint64_t n, uint32_t frame, void *stream)
{
	if (!n) return 0;
	hipLaunchKernelGGL(stamp_kernel, dim3(sgrid(n)), dim3(256), 0, (hipStream_t)stream, ids, lin, n, frame);
	HIPCHK(hipGetLastError());
	return 0;
}
__global__ void __launch_bounds__(256) tile_ranges_kernel(uint32_t *ranges, const uint32_t *ids, const TileRangeGeom g0, int halves)
{
	TileRangeGeom g = g0;
	const int half = blockIdx.x / g.ntiles, tile = blockIdx.x - half * g.ntiles;
	if (halves == 2) { g.row_start = half; g.row_step = 2; }
	uint32_t lo = 0xFFFFFFFFu, hi = 0;
	const long long items = (long long)g.nrows * g.K;
	for (long long it = threadIdx.x; it < items; it += 256) tile_range_item(g, ids, tile, it, lo, hi);
	for (int off = 32; off > 0; off >>= 1) {
		const uint32_t l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
		lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
	}
	__shared__ uint32_t sl[4], sh[4];
	if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = lo; sh[threadIdx.x >> 6] = hi; }
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int i = 1; i < 4; i++) { lo = sl[i] < lo ? sl[i] : lo; hi = sh[i] > hi ? sh[i] : hi; }
		ranges[2 * blockIdx.x] = lo; ranges[2 * blockIdx.x + 1] = hi;
	}
}
int be_scan_tile_ranges(uint32_t *ranges, const uint32_t *ids, TileRangeGeom g, int halves, void *stream)
{
	hipLaunchKernelGGL(tile_ranges_kernel, dim3(g.ntiles * halves), dim3(256), 0, (hipStream_t)stream, ranges, ids, g, halves);
	HIPCHK(hipGetLastError());
	return 0;
}
__global__ void __launch_bounds__(256) tile_eids_kernel(void *eids, const uint32_t *ids, const TileEidGeom g, long long items)
{
	for (long long it = blockIdx.x * 256ll + threadIdx.x; it < items; it += (long long)gridDim.x * 256) tile_eid_item(g, ids, eids, it);
}
int be_scan_tile_eids(void *eids, const uint32_t *ids, TileEidGeom g, void *stream)
{
	const long long items = (long long)g.ntiles * g.N * g.K;
	hipLaunchKernelGGL(tile_eids_kernel, dim3(8192), dim3(256), 0, (hipStream_t)stream, eids, ids, g, items);
	HIPCHK(hipGetLastError());
	return 0;
}
int be_scan_index_to_frame_ids(uint32_t *ids, uint64_t n, uint64_t step, void *stream)
{
	hipLaunchKernelGGL(index_to_frame_kernel, dim3(sgrid(n)), dim3(256), 0, (hipStream_t)stream, ids, n, step);
	HIPCHK(hipGetLastError());
	return 0;
}

// work layout: keys_in | keys_out | vals_in | vals_out | flags | j (n x 4 bytes each), then rocPRIM's temporary storage
static size_t magnitude_temp_bytes(uint64_t n)
{
	size_t a = 0, b = 0;
	float *kf = nullptr; uint32_t *vu = nullptr;
	(void)rocprim::radix_sort_pairs_desc(nullptr, a, kf, kf, vu, vu, n, 0, 32, (hipStream_t)nullptr);
	(void)rocprim::exclusive_scan(nullptr, b, vu, vu, 0u, n, rocprim::plus<uint32_t>(), (hipStream_t)nullptr);
	return (a > b ? a : b) + 256;
}
size_t be_scan_magnitude_work_bytes(uint32_t w, uint32_t h)
{
	const uint64_t n = (uint64_t)w * h;
	return 6 * ((n * 4 + 255) & ~(size_t)255) + magnitude_temp_bytes(n);
}
int be_scan_magnitude_index(uint32_t *idx, const float *coeffs, uint32_t w, uint32_t h, int ch, double q, void *work, size_t work_bytes, uint32_t *limit, void *stream)
{
	const uint64_t n = (uint64_t)w * h;
	if (work_bytes < be_scan_magnitude_work_bytes(w, h)) return -1;
	const size_t slab = (n * 4 + 255) & ~(size_t)255;
	char *base = (char *)work;
	float *k0 = (float *)base, *k1 = (float *)(base + slab);
	uint32_t *v0 = (uint32_t *)(base + 2 * slab), *v1 = (uint32_t *)(base + 3 * slab), *fl = (uint32_t *)(base + 4 * slab), *jj = (uint32_t *)(base + 5 * slab);
	void *temp = base + 6 * slab;
	size_t tb = magnitude_temp_bytes(n);
	hipStream_t st = (hipStream_t)stream;
	hipLaunchKernelGGL(magnitude_keys_kernel, dim3(sgrid(n)), dim3(256), 0, st, k0, v0, coeffs, w, h, ch, q);
	// descending by key; radix sort is stable, so equal keys stay in raster order (the documented tie rule)
	HIPCHK(rocprim::radix_sort_pairs_desc(temp, tb, k0, k1, v0, v1, n, 0, 32, st));
	hipLaunchKernelGGL(magnitude_flags_kernel, dim3(sgrid(n)), dim3(256), 0, st, fl, k1, n);
	HIPCHK(rocprim::exclusive_scan(temp, tb, fl, jj, 0u, n, rocprim::plus<uint32_t>(), st));
	hipLaunchKernelGGL(magnitude_scatter_kernel, dim3(sgrid(n)), dim3(256), 0, st, idx, v1, jj, n);
	HIPCHK(hipGetLastError());
	if (limit) {
		uint32_t last = 0;
		HIPCHK(hipMemcpyAsync(&last, jj + (n - 1), 4, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		*limit = last + 1;
	}
	return 0;
}

}  // namespace dspfft
