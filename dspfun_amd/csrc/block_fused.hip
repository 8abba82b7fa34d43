// block_fused.hip -- kernels of the fused small-block passes (block_core.h) and their dispatch.
#include <hip/hip_runtime.h>
#include "backend.h"
#include "block_core.h"

namespace dspfft {
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

template <int NX, int NY, int NZ, int KIND>
__global__ void __launch_bounds__(BLOCK_THREADS) block_kernel(const BlockArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	float *lds = reinterpret_cast<float *>(lds_raw);
	const int tid = threadIdx.x;
	long long bin, bout;
	int cnt;
	block_base(a, blockIdx.x, bin, bout, cnt);
	block_load_x<NX, NY, NZ, KIND, false>(a, block_axis_args(a.s, 0, NY == 1 && NZ == 1), a.in, nullptr, lds, bin, cnt, tid);
	__syncthreads();
	if constexpr (NY > 1) { block_lines_y<NX, NY, NZ, KIND>(a, block_axis_args(a.s, 1, NZ == 1), lds, cnt, tid); __syncthreads(); }
	if constexpr (NZ > 1) { block_lines_z<NX, NY, NZ, KIND>(a, block_axis_args(a.s, 2, true), lds, cnt, tid); __syncthreads(); }
	block_store_rows<NX, NY, NZ>(a, a.out, lds, bout, cnt, tid);
}

// motion's per-block pipeline in one pass: load (float / 8-bit), REDFT10 along x, y, z, filter, REDFT01 along z, y, x, store
template <int NX, int NY, int NZ, bool IN8, bool OUT8>
__global__ void __launch_bounds__(BLOCK_THREADS) block_roundtrip_kernel(const BlockRtArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	__shared__ unsigned int wg_coded;
	float *lds = reinterpret_cast<float *>(lds_raw);
	const int tid = threadIdx.x;
	if (tid == 0) wg_coded = 0;
	long long bin, bout;
	int cnt;
	block_base(a, blockIdx.x, bin, bout, cnt);
	block_load_x<NX, NY, NZ, KIND_REDFT10, IN8>(a, block_axis_args(a.f, 0, NY == 1 && NZ == 1), a.in, a.in8, lds, bin, cnt, tid);
	__syncthreads();
	// the last forward axis, the filter and the same axis of the inverse run on one line in registers (block_lines_mid)
	unsigned long long mine = 0;
	if constexpr (NZ > 1) {
		if constexpr (NY > 1) { block_lines_y<NX, NY, NZ, KIND_REDFT10>(a, block_axis_args(a.f, 1, false), lds, cnt, tid); __syncthreads(); }
		block_lines_mid<NX, NY, NZ, true>(a, block_axis_args(a.f, 2, true), block_axis_args(a.i, 2, false), a.filt, lds, cnt, tid, mine);
		__syncthreads();
		if constexpr (NY > 1) { block_lines_y<NX, NY, NZ, KIND_REDFT01>(a, block_axis_args(a.i, 1, false), lds, cnt, tid); __syncthreads(); }
	} else {
		block_lines_mid<NX, NY, NZ, false>(a, block_axis_args(a.f, 1, true), block_axis_args(a.i, 1, false), a.filt, lds, cnt, tid, mine);
		__syncthreads();
	}
	if (a.filt.enabled && a.coded) {
		unsigned int m = (unsigned int)mine;
		for (int off = 32; off > 0; off >>= 1) m += __shfl_xor(m, off);
		if ((tid & 63) == 0 && m) atomicAdd(&wg_coded, m);
		__syncthreads();
		if (tid == 0 && wg_coded) atomicAdd(a.coded, (unsigned long long)wg_coded);
	}
	// the inverse's global scale rides on its x pass, the last one here
	block_store_x<NX, NY, NZ, KIND_REDFT01, OUT8>(a, block_axis_args(a.i, 0, true), a.out, a.out8, a.mul8, lds, bout, cnt, tid);
}

template <class K>
static int allow64k(K kern)
{
	return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
}
template <int NX, int NY, int NZ, int KIND>
static int launch_block(const BlockArgs &a, int nwg, size_t lds, void *stream)
{
	static int attr = allow64k(block_kernel<NX, NY, NZ, KIND>);
	if (attr) return attr;
	hipLaunchKernelGGL((block_kernel<NX, NY, NZ, KIND>), dim3(nwg), dim3(BLOCK_THREADS), lds, (hipStream_t)stream, a);
	HIPCHK(hipGetLastError());
	return 0;
}
template <int NX, int NY, int NZ, bool IN8, bool OUT8>
static int launch_block_rt(const BlockRtArgs &a, int nwg, size_t lds, void *stream)
{
	static int attr = allow64k(block_roundtrip_kernel<NX, NY, NZ, IN8, OUT8>);
	if (attr) return attr;
	hipLaunchKernelGGL((block_roundtrip_kernel<NX, NY, NZ, IN8, OUT8>), dim3(nwg), dim3(BLOCK_THREADS), lds, (hipStream_t)stream, a);
	HIPCHK(hipGetLastError());
	return 0;
}

bool be_block_supported(int nx, int ny, int nz)
{
#define DSP_BLOCK_HAS(X_, Y_, Z_) if (nx == X_ && ny == Y_ && nz == Z_) return true;
	DSPFFT_BLOCK_SHAPES(DSP_BLOCK_HAS)
#undef DSP_BLOCK_HAS
	return false;
}

int be_launch_block(const BlockArgs &a, int nwg, size_t lds, void *stream)
{
#define DSP_BLOCK_CASE(X_, Y_, Z_) \
	if (a.nx == X_ && a.ny == Y_ && a.nz == Z_) \
		return a.kind == KIND_REDFT10 ? launch_block<X_, Y_, Z_, KIND_REDFT10>(a, nwg, lds, stream) : launch_block<X_, Y_, Z_, KIND_REDFT01>(a, nwg, lds, stream);
	DSPFFT_BLOCK_SHAPES(DSP_BLOCK_CASE)
#undef DSP_BLOCK_CASE
	return -1;
}

int be_launch_block_roundtrip(const BlockRtArgs &a, int nwg, size_t lds, void *stream)
{
#define DSP_BLOCK_CASE(X_, Y_, Z_) \
	if (a.nx == X_ && a.ny == Y_ && a.nz == Z_) { \
		if (a.in8) return a.out8 ? launch_block_rt<X_, Y_, Z_, true, true>(a, nwg, lds, stream) : launch_block_rt<X_, Y_, Z_, true, false>(a, nwg, lds, stream); \
		return a.out8 ? launch_block_rt<X_, Y_, Z_, false, true>(a, nwg, lds, stream) : launch_block_rt<X_, Y_, Z_, false, false>(a, nwg, lds, stream); \
	}
	DSPFFT_BLOCK_SHAPES(DSP_BLOCK_CASE)
#undef DSP_BLOCK_CASE
	return -1;
}
}  // namespace dspfft
