// spec_fused.h -- device-side pieces of the fused column roundtrip shared by spec_kernels.h and jit_kernels.h.
#pragma once
#include <hip/hip_runtime.h>
#include "dct_spec.h"
#ifndef DSP_STAMP
#define DSP_STAMP(i) ((void)0)
#endif
#include "motion_filter.h"

namespace dspfft {

// forward REDFT10 -> motion filter -> inverse REDFT01 along the tile's axis in one launch: the tile is read once and
// written once instead of three times each (forward store + filter read/write + inverse load saved)
struct FilterOp {
	MotionFilter p;
	__device__ SigVec<float, 2> operator()(long long e, SigVec<float, 2> v, unsigned long long &coded) const
	{
		if (!p.enabled) return v;
		float4 f; f.x = v.s[0].x; f.y = v.s[0].y; f.z = v.s[1].x; f.w = v.s[1].y;
		f = motion_filter4(p, (uint32_t)e, f, coded);
		v.s[0].x = f.x; v.s[0].y = f.y; v.s[1].x = f.z; v.s[1].y = f.w;
		return v;
	}
};

// waves per SIMD to ask of the register allocator so that as many workgroups stay resident as the tile's LDS allows
// (capped at 4 = 128 VGPRs): without the cap the allocator spends the whole budget and one workgroup fills a CU
template <class S> constexpr int rt_waves_per_simd()
{
	const int wgs = (int)((160 * 1024) / S::LDS), w = wgs * S::T / 256;
	return w < 1 ? 1 : w > 4 ? 4 : w;
}

// the body of the fused column roundtrip, shared by the listed kernels (spec_kernels.h, dynamic LDS) and the ones compiled at plan
// time (jit_kernels.h, static LDS)
// the fused scan step's and zoom's fields pinned to "off" on a local copy of the arguments: every branch on them folds away after inlining
template <class PA> __host__ __device__ inline PA plain_args(const PA &a_)
{
	PA a = a_;
	a.mask = nullptr; a.zflags = nullptr; a.zranges = nullptr; a.accumulate = 0; a.win_lo = a.win_hi = 0; a.alt_out = 0; a.in_mul = nullptr; a.in_rev = 0;
	// (lean_off stays a run-time value: DSPFFT_LEAN01=0 must reach the plain kernels too)
	return a;
}
template <class PA> static inline bool is_plain(const PA &a) { return !a.mask && !a.zflags && !a.accumulate && a.win_hi <= 0 && !a.alt_out && !a.in_mul && !a.in_rev; }

template <class S>
__device__ __forceinline__ void col_roundtrip_body(unsigned char *lds, const typename S::PA &af_, const typename S::PA &ai_, const FilterOp &filt, unsigned long long *coded)
{
	// motion's roundtrip is a plain pair of passes: the fused scan step's mask / flags / accumulation and zoom's window / modulation / alternating
	// sign are pinned off (launch_col_roundtrip refuses them), so their per-load selects and the masked variants of prefetch fold away
	const typename S::PA af = plain_args(af_), ai = plain_args(ai_);
	__shared__ unsigned int wg_coded;       // non-zero quantised coefficients of this tile (one global atomic per workgroup)
	typename S::V *buf = reinterpret_cast<typename S::V *>(lds);
	const int tid = threadIdx.x;
	if (tid == 0) wg_coded = 0;
	typename S::StateRT st;
	long long bin, bout;
	S::base(af, blockIdx.x, bin, bout);
	bool hit = false;
	DSP_STAMP(0);
	S::template prefetch<KIND_REDFT10>(af, bin, tid, st, hit);
	S::template phase<KIND_REDFT10, 0>(af, buf, bout, tid, st);
	__syncthreads();
	DSP_STAMP(1);
	// The empty asm statements make the thread index (and, below, the inverse plan's table pointers) opaque at each
	// phase: otherwise index arithmetic and twiddle loads of LATER phases are hoisted to the top of the kernel and
	// stay live across every barrier (measured: 176 VGPRs -> 1 workgroup per CU; with them: <= 128).
	static_for<1, S::NS + 2>([&](auto ph) {
		int t = tid; asm volatile("" : "+v"(t));
		S::template phase<KIND_REDFT10, ph>(af, buf, bout, t, st);
		__syncthreads();
		DSP_STAMP(1 + ph);
	});
	unsigned long long mine = 0;
	int t = tid; asm volatile("" : "+v"(t));
	S::mid_read(af, ai, buf, bout, t, st, filt, mine);
	if (coded) {
		unsigned int m = (unsigned int)mine;
		for (int off = 32; off > 0; off >>= 1) m += __shfl_xor(m, off);
		if ((tid & 63) == 0 && m) atomicAdd(&wg_coded, m);
	}
	__syncthreads();
	DSP_STAMP(10);
	if (coded && tid == 0 && wg_coded) atomicAdd(coded, (unsigned long long)wg_coded);
	asm volatile("" : "+v"(t));
	S::mid_write(buf, t, st);
	__syncthreads();
	DSP_STAMP(11);
	typename S::PA a2 = ai;
	asm volatile("" : "+s"(a2.W), "+s"(a2.T), "+s"(a2.out));
	static_for<1, S::NPH>([&](auto ph) {
		asm volatile("" : "+v"(t));
		S::template phase<KIND_REDFT01, ph>(a2, buf, bout, t, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
		DSP_STAMP(12 + ph);
	});
}

}  // namespace dspfft
