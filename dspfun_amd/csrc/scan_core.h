// scan_core.h -- the integer scan orders of the reference's `scan` tool other than zigzag (scan/scan_methods.c:59-67,122-208,
// 298-331), as per-element functions shared by the HIP kernels (scan_methods.hip) and the test-only CPU emulation.
// Two views of a method:
//   owner index  -- the scan index that yields pixel (y, x), for methods where every pixel belongs to exactly one index
//                   (all but `box`); frame of the pixel = owner index / step (scan/scan.c:421-427)
//   coordinates  -- the j-th coordinate of scan index i as a linear offset y*w + x, for the closed-form methods (all but
//                   radial / iradial, whose buckets are only defined per pixel), including the reference's quirks:
//                   `box` emits x = i unclamped on its first leg (the tool's pointer arithmetic then lands in a later row;
//                   offsets past the image are reported as SCAN_NONE) and repeats row h-1 for every i >= h-1 on wide images;
//                   `ibox` emits its corner twice.
#pragma once
#include <math.h>
#include <stdint.h>
#include "radix.h"
#include "dct_core.h"

namespace dspfft {

enum ScanMethod {       // numbering of host/scan_orders.h
	SCANM_HORIZONTAL = 0, SCANM_VERTICAL, SCANM_ZIGZAG, SCANM_ROW, SCANM_COLUMN, SCANM_DIAGONAL, SCANM_MIRROR, SCANM_BOX, SCANM_IBOX,
	SCANM_RADIAL, SCANM_IRADIAL, SCANM_COUNT
};
constexpr uint32_t SCAN_NONE = 0xffffffffu;

// rint(hypot(x, y)) for integers, exactly (scan_methods.c:298-331 with the default rounding): no tie can occur because
// (k + 1/2)^2 is never an integer, so the result is k = floor(sqrt(s)) when s <= k^2 + k and k + 1 otherwise
DSP_HD uint64_t rint_hypot(uint64_t x, uint64_t y)
{
	const uint64_t s = x * x + y * y;
	uint64_t k = (uint64_t)sqrt((double)s);
	while (k * k > s) k--;
	while ((k + 1) * (k + 1) <= s) k++;
	return s - k * k > k ? k + 1 : k;
}

DSP_HD uint64_t scan_umin(uint64_t a, uint64_t b) { return a < b ? a : b; }

// scan index that owns pixel (y, x); SCANM_BOX has no single owner and SCANM_ZIGZAG is generated index-first (zigzag_lin)
DSP_HD uint64_t scan_owner_index(int method, uint64_t w, uint64_t h, uint64_t y, uint64_t x)
{
	switch (method) {
	case SCANM_HORIZONTAL: return y * w + x;                        // scan_methods.c:59-62
	case SCANM_VERTICAL: return x * h + y;                          // :64-67
	case SCANM_ROW: return y;                                       // :146-151
	case SCANM_COLUMN: return x;                                    // :153-158
	case SCANM_DIAGONAL: return x + y;                              // :160-165
	case SCANM_MIRROR: return x > y ? x - y : y - x;                // :167-187
	case SCANM_IBOX: return scan_umin(x, y);                        // :135-144
	case SCANM_RADIAL: return rint_hypot(x, y);                     // :298-331
	case SCANM_IRADIAL: return rint_hypot(w - 1, h - 1) + 1 - rint_hypot(w - x - 1, h - y - 1) - 1;
	default: return 0;
	}
}

// number of coordinates scan index i yields (closed-form methods)
DSP_HD uint64_t scan_interval(int method, uint64_t w, uint64_t h, uint64_t i)
{
	switch (method) {
	case SCANM_ROW: return w;
	case SCANM_COLUMN: return h;
	case SCANM_DIAGONAL: { const uint64_t y0 = i < h ? i : h - 1, x0 = i - y0; return x0 < w ? scan_umin(y0 + 1, w - x0) : 0; }
	case SCANM_MIRROR:
		if (i == 0) return scan_umin(w, h);
		return (i < w ? scan_umin(h, w - i) : 0) + (i < h ? scan_umin(w, h - i) : 0);
	case SCANM_BOX: return (i < h ? i : h - 1) + (i < w ? i : w - 1) + 1;
	case SCANM_IBOX: return (w - i) + (h - i);
	default: return 1;
	}
}

// linear offset of the j-th coordinate of scan index i (j < scan_interval), or SCAN_NONE when it falls outside the image
DSP_HD uint32_t scan_coord_lin(int method, uint64_t w, uint64_t h, uint64_t i, uint64_t j)
{
	uint64_t y = 0, x = 0;
	switch (method) {
	case SCANM_HORIZONTAL: y = i / w; x = i % w; break;
	case SCANM_VERTICAL: y = i % h; x = i / h; break;
	case SCANM_ROW: y = i; x = j; break;
	case SCANM_COLUMN: y = j; x = i; break;
	case SCANM_DIAGONAL: { const uint64_t y0 = i < h ? i : h - 1; y = y0 - j; x = i - y0 + j; break; }
	case SCANM_MIRROR:
		if (i == 0) { y = x = j; break; }
		{
			const uint64_t n1 = i < w ? scan_umin(h, w - i) : 0;
			if (j < n1) { const uint64_t xx = n1 - j; y = xx - 1; x = xx + i - 1; }
			else { const uint64_t n2 = scan_umin(w, h - i), yy = n2 - (j - n1); y = yy + i - 1; x = yy - 1; }
		}
		break;
	case SCANM_BOX: {
		const uint64_t ymax = i < h ? i : h - 1;
		if (j < ymax) { y = j; x = i; }             // x = i is NOT clamped (scan_methods.c:126)
		else { y = ymax; x = j - ymax; }
		break;
	}
	case SCANM_IBOX:
		if (j < w - i) { y = i; x = i + j; } else { y = i + (j - (w - i)); x = i; }
		break;
	default: return SCAN_NONE;
	}
	const uint64_t lin = y * w + x;
	return lin < w * h ? (uint32_t)lin : SCAN_NONE;
}

// magnitude (scan_methods.c:240-296): the sort key of pixel (y, x), exactly as the COEFF_PRECISION=F / INTERMEDIATE_PRECISION=D build
// computes it: a double sum of |c_z|, the per-index normalisation in double, optional quantisation, stored as float
DSP_HD float scan_magnitude_key(const float *c, int channels, uint64_t y, uint64_t x, double qfactor)
{
	double sum = 0;
	for (int z = 0; z < channels; z++) sum += (double)fabsf(c[z]);
	const double P = 1.41421356237309504880;          // M_SQRT2 = precision.h:130 P_SQRT2i for intermediate = double
	const double norm = (x ? P : 1.0) * (y ? P : 1.0);
	return (float)(qfactor != 0.0 ? rint(sum * norm * qfactor / channels) : sum * norm);
}

// (min, max) owner id over the elements of one column tile (sparse scan frames, PassGeom::zranges); ids of 0xFFFFFFFF (the DC pixel,
// scan.c:377-383) are no frame's.  Rows row_start, row_start + row_step, ...: a half tile of a split column pass takes every other row.
struct TileRangeGeom { int K, ntiles, nrows, row_start, row_step; long long es; FastDiv div; };
DSP_HD void tile_range_item(const TileRangeGeom &g, const uint32_t *ids, int tile, long long item, uint32_t &lo, uint32_t &hi)
{
	const long long r = item / g.K;
	const int j = (int)(item - r * g.K);
	const uint32_t id = ids[g.div.div((uint32_t)((g.row_start + r * g.row_step) * g.es + (long long)tile * g.K + j))];
	if (id != 0xFFFFFFFFu) { lo = id < lo ? id : lo; hi = id > hi ? id : hi; }
}

// The owner id of every element of the image in the order the column tiles of a fused scan step read them (dct_spec.h, eid_index:
// [tile][row][K], the rows of a split pass by parity), `bytes` (1 or 2) per id; ids that do not fit, and 0xFFFFFFFF, become all ones.
struct TileEidGeom { int K, ntiles, N, halves, bytes; long long es; FastDiv div; };
DSP_HD void tile_eid_item(const TileEidGeom &g, const uint32_t *ids, void *eids, long long item)
{
	const long long per_tile = (long long)g.N * g.K;
	const int t = (int)(item / per_tile);
	const long long rem = item - (long long)t * per_tile;
	const int r = (int)(rem / g.K), k = (int)(rem - (long long)r * g.K);
	const int y = g.halves == 2 ? (r < g.N / 2 ? 2 * r : 2 * (r - g.N / 2) + 1) : r;       // the inverse of eid_index's row order
	const uint32_t id = ids[g.div.div((uint32_t)((long long)y * g.es + (long long)t * g.K + k))];
	if (g.bytes == 1) reinterpret_cast<uint8_t *>(eids)[item] = id < 0xffu ? (uint8_t)id : (uint8_t)0xffu;
	else reinterpret_cast<uint16_t *>(eids)[item] = id < 0xffffu ? (uint16_t)id : (uint16_t)0xffffu;
}

}  // namespace dspfft
