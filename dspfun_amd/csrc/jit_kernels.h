// jit_kernels.h -- the two kernels a plan compiles AT PLAN TIME (hiprtc, backend_hip.hip be_jit_build) for a frame size that has no
// entry in spec_list.h: the same phases as row_spec_kernel / col_spec_kernel (spec_kernels.h) over a RowSpecT / ColSpecT chosen by the
// planner (engine.cpp jit_choose), with the LDS tile as a static array so that no launch attribute is needed.  Device code only.
#pragma once
#include "dct_spec.h"

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

}  // namespace dspfft
