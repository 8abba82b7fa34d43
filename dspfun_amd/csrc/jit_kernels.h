// jit_kernels.h -- the two kernels a plan compiles AT PLAN TIME (hiprtc, backend_hip.hip be_jit_build) for a frame size that has no
// entry in spec_list.h: the same phases as row_spec_kernel / col_spec_kernel (spec_kernels.h) over a RowSpecT / ColSpecT chosen by the
// planner (engine.cpp jit_choose), with the LDS tile as a static array so that no launch attribute is needed.  Device code only.
#pragma once
#include "dct_spec.h"
#include "spec_fused.h"

namespace dspfft {

template <class S, int KIND>
__global__ void __launch_bounds__(S::T, S::WPE) jit_row(const typename S::PA a)
{
	__shared__ __attribute__((aligned(32))) unsigned char lds[S::LDS];
	typename S::CX *planes = reinterpret_cast<typename S::CX *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	row_base(a, blockIdx.x, bin, bout);
	S::template prefetch<KIND>(a, bin, tid, st);
	S::template phase<KIND, 0>(a, planes, bout, tid, st);
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, planes, bout, tid, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
	});
}

template <class S, int KIND>
__global__ void __launch_bounds__(S::T, S::WPE) jit_col(const typename S::PA a)
{
	__shared__ __attribute__((aligned(32))) unsigned char lds[S::LDS];
	typename S::V *buf = reinterpret_cast<typename S::V *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	bool hit = false;
	S::base(a, blockIdx.x, bin, bout);
	S::template prefetch<KIND>(a, bin, tid, st, hit);
	S::template phase<KIND, 0>(a, buf, bout, tid, st);
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, buf, bout, tid, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
	});
}

// planar rows reading 8-bit samples (REDFT10) or writing quantised 8-bit samples (REDFT01): row_spec_u8_kernel's twin
template <class S, int KIND>
__global__ void __launch_bounds__(S::T, S::WPE) jit_row_u8(const typename S::PA a_, const U8IO io)
{
	const typename S::PA a = plain_args(a_);
	__shared__ __attribute__((aligned(32))) unsigned char lds[S::LDS];
	typename S::CX *planes = reinterpret_cast<typename S::CX *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	row_base(a, blockIdx.x, bin, bout);
	S::template prefetch<KIND>(a, bin, tid, st, &io);
	S::template phase<KIND, 0>(a, planes, bout, tid, st, &io);
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, planes, bout, tid, st, &io);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
	});
}

// forward -> filter -> inverse along the tile's axis in one launch: col_roundtrip_kernel's twin (spec_fused.h)
template <class S>
__global__ void __launch_bounds__(S::T, rt_waves_per_simd<S>()) jit_col_rt(const typename S::PA af, const typename S::PA ai, const FilterOp filt, unsigned long long *coded)
{
	__shared__ __attribute__((aligned(32))) unsigned char lds[S::LDS];
	col_roundtrip_body<S>(lds, af, ai, filt, coded);
}

}  // namespace dspfft
