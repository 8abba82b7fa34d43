// motion_filter.h -- motion's coefficient-domain filter (motion/motion.c:683-744) as one pointwise function of
// (coefficient position, value), shared by the standalone kernel (pointwise.hip: dspfft_motion_filter) and by the
// fused forward -> filter -> inverse column pass (dct_spec.h: ColSpec::mid_*, dspfft_execute_roundtrip).
#pragma once
#include <math.h>
#include "dct_core.h"

namespace dspfft {

struct MotionFilter {
	int ad, ah, aw, mh, mw;                  // active block and the buffer's row/plane extents
	int b0d, b0h, b0w, b1d, b1h, b1w;        // pass band [begin, end) per axis
	float damp, boost, thr_lo, thr_hi;
	int preserve_dc;                         // 0 none, 1 dc, 2 grey
	float grey_add, quantizer;
	// position from an element offset into the buffer: x = e % mw, y = (e / mw) % mh, z = (e / (mw mh)) % bd
	FastDiv div_mw, div_mh, div_bd;
	int bd;                                  // frames per block as laid out (the embedding depth; 1 = every frame its own block)
	int enabled;
	int quant_only;                          // nothing but the quantiser acts (no damp/boost/threshold/DC rule): positions are not needed
	float rquant = 0.f;                      // 1 / quantizer rounded to float (motion_filter_set_divs), or 0: divide
};

inline FastDiv motion_filter_div(uint32_t d)
{
	FastDiv f; f.d = d; f.mul = d >= 2 ? (uint32_t)((((uint64_t)1 << 32) + d - 1) / d) : 0; return f;
}
inline void motion_filter_set_divs(MotionFilter &p, int block_depth)
{
	p.bd = block_depth;
	p.quant_only = p.quantizer > 0.f && p.damp == 1.f && p.boost == 1.f && !(p.thr_hi > 0.f) && (p.preserve_dc == 0 || !(p.b0d || p.b0h || p.b0w));
	p.rquant = (p.quantizer > 1e-30f && p.quantizer < 1e30f) ? 1.0f / p.quantizer : 0.f;
	p.div_mw = motion_filter_div((uint32_t)p.mw); p.div_mh = motion_filter_div((uint32_t)p.mh); p.div_bd = motion_filter_div((uint32_t)block_depth);
}

// motion.c:744  coeffs = mi(round)(coeffs / quantizer) * quantizer  with coeffs and quantizer of type `coeff` (float in motion's build,
// motion/Makefile:1): the QUOTIENT is a float division, rounded to float before round() sees it; round() of a float is exact in any wider
// type, and the product of an integer below 2^24 with a float is exact in `intermediate`, so the assignment rounds once -- exactly what a float
// multiplication does.  All in single precision, bit for bit (tests/test_ref_motion.py against the reference's compiled lines).  Round 4 divided
// in double, which rounds differently when the double quotient and its float rounding lie on opposite sides of a half-integer, and cost four
// times the instructions (v_div_scale_f64 ...).
// The float quotient itself, without the divider: with rq = RN(1 / q), t0 = RN(v rq) is within a couple of units in the last place of v / q;
// each step t' = RN(t + (v - q t) rq) -- the residual v - q t exact in one FMA -- brings an approximation that is within one unit to the
// correctly rounded quotient (Markstein's theorem, for a correctly rounded reciprocal), so two steps give RN(v / q): 5 full-rate instructions
// for the hardware sequence's 10 and its quarter-rate v_rcp_f32 (checked against the division on 2^26 values per quantiser, adversarial
// significands included: tests/test_ref_motion.py).  rq == 0 (quantisers outside [1e-30, 1e30]): divide.
// Most quotients are nowhere near a rounding boundary: t0 is within 1.5 x 2^-23 |t0| of RN(v / q), so unless its distance to the nearest
// half-integer is within 4 x 2^-23 |t0| both round to the same integer, which is then not a tie either (round-to-even = roundf): 6 instructions
// and a skipped branch instead of 12.  (|t0| >= 2^20 always takes the exact path.)
DSP_HD float motion_quantise(float v, float q, float rq)
{
	if (rq == 0.f) return roundf(v / q) * q;
	const float t0 = v * rq;
	float r = rintf(t0);
	const float d = t0 - r;                                                    // exact
	if (fabsf(fabsf(d) - 0.5f) <= fabsf(t0) * 0x1p-21f) {
		const float t1 = fmaf(fmaf(-t0, q, v), rq, t0);
		r = roundf(fmaf(fmaf(-t1, q, v), rq, t1));
	}
	return r * q;
}

// one coefficient at block position (z, y, x); `coded` counts the non-zero quantised coefficients (motion.c:743)
DSP_HD float motion_filter_at(const MotionFilter &p, int z, int y, int x, float v, unsigned long long &coded)
{
	const float dc = v;                                                        // motion.c:650 (element 0 only)
	const bool inside = z >= p.b0d && z < p.b1d && y >= p.b0h && y < p.b1h && x >= p.b0w && x < p.b1w;
	if (!inside) { if (p.damp != 1.f) v *= p.damp; }                          // :683-714 the six face slabs = the complement of the box
	else if (p.boost != 1.f) v *= p.boost;                                    // :715-719
	if (p.thr_hi > 0.f) { const float a = fabsf(v); if (a < p.thr_lo || a > p.thr_hi) v = 0.f; }   // :721-728
	if (x == 0 && y == 0 && z == 0 && p.preserve_dc) {                        // :730-738
		const bool dcstop = p.b0d || p.b0h || p.b0w;
		if (dcstop || p.boost != 1.f || p.thr_hi > 0.f) {
			if (p.preserve_dc == 1) v = dc;
			else v += p.grey_add;
		}
	}
	if (p.quantizer > 0.f) { v = motion_quantise(v, p.quantizer, p.rquant); coded += (v != 0.f); }   // :740-744
	return v;
}

// one element by offset; elements outside the active block (embedding padding) are returned unchanged
DSP_HD float motion_filter_elem(const MotionFilter &p, uint32_t e, float v, unsigned long long &coded)
{
	const uint32_t row = p.div_mw.div_exact(e), x = e - row * (uint32_t)p.mw;
	const uint32_t pl = p.div_mh.div_exact(row), y = row - pl * (uint32_t)p.mh;
	const uint32_t z = pl - p.div_bd.div_exact(pl) * (uint32_t)p.bd;
	if ((int)x >= p.aw || (int)y >= p.ah || (int)z >= p.ad) return v;
	return motion_filter_at(p, (int)z, (int)y, (int)x, v, coded);
}

// four consecutive elements starting at element offset e (< 2^31) of the working buffer
DSP_HD float4 motion_filter4(const MotionFilter &p, uint32_t e, float4 v, unsigned long long &coded)
{
	if (p.quant_only) {
		float r[4] = {v.x, v.y, v.z, v.w};
		unsigned int nz = 0;
		for (int q = 0; q < 4; q++) {
			r[q] = motion_quantise(r[q], p.quantizer, p.rquant);                                       // motion.c:740-744
			nz += (r[q] != 0.f);
			DSP_SCHED_FENCE();
		}
		coded += nz;
		float4 o; o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
		return o;
	}
	const uint32_t row = p.div_mw.div_exact(e);
	int x = (int)(e - row * (uint32_t)p.mw);
	const uint32_t pl = p.div_mh.div_exact(row);
	int y = (int)(row - pl * (uint32_t)p.mh);
	int z = (int)(pl - p.div_bd.div_exact(pl) * (uint32_t)p.bd);
	float r[4] = {v.x, v.y, v.z, v.w};
	for (int q = 0; q < 4; q++) {
		r[q] = motion_filter_at(p, z, y, x, r[q], coded);
		if (++x == p.mw) { x = 0; if (++y == p.mh) { y = 0; if (++z == p.bd) z = 0; } }
		DSP_SCHED_FENCE();
	}
	float4 o; o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
	return o;
}

// ---- the two ends of motion's block loop as functions of one sample (motion_ops.hip's kernels; pinned to the reference's compiled lines through the
// test-only emulation library: tests/test_ref_motion.py).  Scalar math in double (`intermediate` is long double in motion's build, motion/Makefile:2).
enum { MOTION_MODE_NONE = 0, MOTION_MODE_ABS = 1, MOTION_MODE_SHIFT = 2, MOTION_MODE_FLAT = 3, MOTION_MODE_COPY = 4 };   // = DSPFFT_MOTION_* (include/dspfft.h)
// motion.c:621-637: pel = the 8-bit sample, or the float sample * 255 (float_pixels, :623); the --ispec decode (:627-635); the value stored as a coefficient
DSP_HD double motion_load_pel(double pel, int mode, double ic, double norm)
{
	switch (mode) {
	case MOTION_MODE_SHIFT: return copysign(expm1(fabs((pel - 127.5) / ic)), pel - 127.5) / norm;
	case MOTION_MODE_FLAT: return (pel - 127.5) * 2 / norm / norm;
	case MOTION_MODE_COPY: return pel / norm / norm;
	default: return pel;
	}
}
// motion.c:759-771: the output pel of a coefficient before it is stored (8-bit: clamp + lround, :776; float_pixels: / 255, :774)
DSP_HD double motion_store_pel(double c, int mode, double scalefactor, double norm, double cc)
{
	double pel = c * scalefactor * norm;
	switch (mode) {
	case MOTION_MODE_ABS: return cc * log1p(fabs(pel));
	case MOTION_MODE_SHIFT: return cc * copysign(log1p(fabs(pel)), pel) + 127.5;
	case MOTION_MODE_FLAT: return pel * norm / 2 + 127.5;
	default: return pel * norm;
	}
}

}  // namespace dspfft
