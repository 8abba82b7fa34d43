// elementwise_core.h -- per-element arithmetic of the device-side helpers in include/dspfft.h,
// shared between the HIP kernels and the test-only CPU emulation.
#pragma once
#include <math.h>
#include "radix.h"

namespace dspfft {

// Largest d with d(d+1)/2 <= i.  The reference computes (size_t)(sqrt(i*2+0.25)-0.5) in double
// (scan/scan_methods.c:69-71); the estimate is corrected with exact integer steps so the result
// does not depend on how the device rounds sqrt.
DSP_HD uint64_t tri_floor_exact(uint64_t i)
{
	uint64_t d = (uint64_t)(sqrt((double)(2 * i) + 0.25) - 0.5);
	while ((d + 1) * (d + 2) / 2 <= i) d++;
	while (d * (d + 1) / 2 > i) d--;
	return d;
}

// scan/scan_methods.c:77-115 (scan_zigzag) -> linear offset y*w+x of scan index i
DSP_HD uint32_t zigzag_lin(uint32_t w32, uint32_t h32, uint64_t i)
{
	const uint64_t w = w32, h = h32;
	const uint64_t m = w < h ? w : h, head = m * (m + 1) / 2, area = w * h;
	uint64_t y, x;
	if (i < head) {
		const uint64_t d = tri_floor_exact(i);
		uint64_t r = i - d * (d + 1) / 2;
		if ((d & 1) == 0) r = d - r;
		y = r; x = d - r;
	} else if (area - i <= head) {
		const uint64_t j = area - i - 1, d = tri_floor_exact(j);
		uint64_t r = j - d * (d + 1) / 2;
		if ((((w + h - 1) - d - 1) & 1) == 0) r = d - r;
		y = (h - 1) - r; x = (w - 1) - (d - r);
	} else {
		const uint64_t band = (i - head) / m;
		uint64_t r = m - (i - (band * m + head));
		if (((band + m) & 1) == 0) r = m - r + 1;
		if (w < h) { r = m - r + 1; y = band + r; x = w - r; }
		else { y = h - r; x = band + r; }
	}
	return (uint32_t)(y * w + x);
}

// motion/motion.c:776: pel > 255 ? 255 : pel < 0 ? 0 : lround(pel)   (intermediate = double)
DSP_HD uint8_t quantise_u8(double pel)
{
	if (pel > 255.0) return 255;
	if (pel < 0.0) return 0;
	return (uint8_t)round(pel);
}
// quantise_u8((double)v * mul) for the kernels that store a quantised pixel per sample (rows' 8-bit end, block transforms): the float product
// v * (float)mul is within 2^-23 |p| of the double one, so unless it lies within 2^-21 |p| of a half-integer both round to the same integer
// (and clamp alike: 255.5 and -0.5 are such half-integers) -- six single-precision instructions, where the double path is a conversion, a
// multiplication, three comparisons, two 64-bit selects, an addition and a conversion back; next to a boundary the double path decides.
// NaN -> 0 and +-inf -> 255 / 0, as the double path's conversions give on the device.
DSP_HD uint32_t quantise_u8_of(float v, double mul, float mulf)
{
	const float p = v * mulf;
	const float r = rintf(p);
	const float d = p - r;                                                     // exact
	if (fabsf(fabsf(d) - 0.5f) <= fabsf(p) * 0x1p-21f) return quantise_u8((double)v * mul);
	return (uint32_t)(int)fminf(fmaxf(r, 0.f), 255.f);
}

}  // namespace dspfft
