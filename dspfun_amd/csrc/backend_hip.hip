// backend_hip.hip -- gfx950 (MI355X / CDNA4) kernels and launches behind backend.h.
//
// Generic (runtime-geometry) kernels: one workgroup per line (ROW), per column tile (COL) or per
// signal (DENSE); the whole signal set of the workgroup stays in LDS between global load and global
// store, so each pass moves exactly one read + one write of the array through HBM.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "backend.h"
#include "dct_spec.h"
#include "spec_list.h"
#include "elementwise_core.h"
#include "spec_kernels.h"

namespace dspfft {

// the specialised kernels are instantiated in spec_inst_*.hip (compiled in parallel); here they are only referenced
#define DSP_EXTERN_ROW(N, C, T, ...) \
	extern template int launch_row_spec<RowSpec<N, C, T, __VA_ARGS__>, 0>(const PassArgs &, int, void *); \
	extern template int launch_row_spec<RowSpec<N, C, T, __VA_ARGS__>, 1>(const PassArgs &, int, void *); \
	extern template int launch_row_sum2<RowSpec<N, C, T, __VA_ARGS__>>(const PassArgs &, const PassArgs &, int, void *); \
	extern template int launch_row_spec_u8<RowSpec<N, C, T, __VA_ARGS__>, 0>(const PassArgs &, const U8IO &, int, void *); \
	extern template int launch_row_spec_u8<RowSpec<N, C, T, __VA_ARGS__>, 1>(const PassArgs &, const U8IO &, int, void *);
#define DSP_EXTERN_COL(N, K, T, ...) \
	extern template int launch_col_spec<ColSpec<N, K, T, __VA_ARGS__>, 0>(const PassArgs &, int, void *); \
	extern template int launch_col_spec<ColSpec<N, K, T, __VA_ARGS__>, 1>(const PassArgs &, int, void *); \
	extern template int launch_col_roundtrip<ColSpec<N, K, T, __VA_ARGS__>>(const PassArgs &, const PassArgs &, const MotionFilter &, unsigned long long *, int, void *);
#define DSP_EXTERN_HALF(N, K, T, ...) \
	extern template int launch_col_half<ColHalfSpec<N, K, T, __VA_ARGS__>, 0>(const PassArgs &, int, void *); \
	extern template int launch_col_half<ColHalfSpec<N, K, T, __VA_ARGS__>, 1>(const PassArgs &, int, void *);
#define DSP_EXTERN_PAIR(N, C, T, ...) \
	extern template int launch_row_pair<RowSpec<N, C, T, __VA_ARGS__>, 0>(const PassArgs &, int, void *); \
	extern template int launch_row_pair<RowSpec<N, C, T, __VA_ARGS__>, 1>(const PassArgs &, int, void *);
#define DSP_EXTERN_ROW_D(N, C, T, ...) \
	extern template int launch_row_spec<RowSpecT<double, N, C, T, __VA_ARGS__>, 0>(const PassArgsD &, int, void *); \
	extern template int launch_row_spec<RowSpecT<double, N, C, T, __VA_ARGS__>, 1>(const PassArgsD &, int, void *);
#define DSP_EXTERN_COL_D(N, K, T, ...) \
	extern template int launch_col_spec<ColSpecT<double, N, K, T, __VA_ARGS__>, 0>(const PassArgsD &, int, void *); \
	extern template int launch_col_spec<ColSpecT<double, N, K, T, __VA_ARGS__>, 1>(const PassArgsD &, int, void *);
#define DSP_EXTERN_ZOOMX(M, T, ...) extern template int launch_zoomx<RowDuoT<M, T, __VA_ARGS__>, 3>(const ZoomXArgs &, int, bool, void *);
DSPFFT_ZOOMX_SPECS(DSP_EXTERN_ZOOMX)
#define DSP_EXTERN_CZT(P, T, ...) \
	extern template int launch_czt_rows<CztSpecT<P, T, __VA_ARGS__>>(const CztArgs &, void *); \
	extern template int launch_czt_spectrum<CztSpecT<P, T, __VA_ARGS__>>(const CztArgs &, cf *, void *);
DSPFFT_CZT_SPECS(DSP_EXTERN_CZT)
DSPFFT_ROW_SPECS_F64(DSP_EXTERN_ROW_D)
DSPFFT_COL_SPECS_F64(DSP_EXTERN_COL_D)
DSPFFT_ROW_SPECS(DSP_EXTERN_ROW)
DSPFFT_COL_SPECS(DSP_EXTERN_COL)
DSPFFT_COL_HALF_SPECS(DSP_EXTERN_HALF)
DSPFFT_ROW_PAIR_SPECS(DSP_EXTERN_PAIR)

// ---------------------------------------------------------------------------------------------
// MAXT: the largest workgroup the instantiation is launched with (512 leaves the register allocator 256 VGPRs, which
// the double-precision radix-16 butterflies need; 1024 caps it at 128)
template <int KIND, class R, int MAXT>
__global__ void __launch_bounds__(MAXT) row_kernel(const PassArgsT<R> a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	__shared__ long long bases[2 * 16];
	cx<R> *buf = reinterpret_cast<cx<R> *>(lds);
	const int tid = threadIdx.x, nthr = blockDim.x, L = a.N / 2, B = a.C * a.LPW;
	const int cnt = row_bases(a, blockIdx.x, bases, tid);
	__syncthreads();
	if (KIND == KIND_REDFT10) row_load10(a, buf, bases, cnt, tid, nthr);
	else row_load01(a, buf, bases, cnt, tid, nthr);
	__syncthreads();
	for (int s = 0; s < a.fft.ns; s++) {
		fft_stage(buf, L, a.fft.st[s], B, a.divB, a.W, tid, nthr);
		__syncthreads();
	}
	if (KIND == KIND_REDFT10) row_post10(a, buf, bases, cnt, tid, nthr);
	else row_store01(a, buf, bases, cnt, tid, nthr);
}

template <int KIND, class R, int MAXT>
__global__ void __launch_bounds__(MAXT) col_kernel(const PassArgsT<R> a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	cx<R> *buf = reinterpret_cast<cx<R> *>(lds);
	const int tid = threadIdx.x, nthr = blockDim.x;
	long long bin, bout; int valid;
	col_base(a, blockIdx.x, bin, bout, valid);
	if (KIND == KIND_REDFT10) col_load2(a, buf, bin, valid, tid, nthr);
	else col_pre3(a, buf, bin, valid, tid, nthr);
	__syncthreads();
	for (int s = 0; s < a.fft.ns; s++) {
		fft_stage(buf, a.N, a.fft.st[s], a.B, a.divB, a.W, tid, nthr);
		__syncthreads();
	}
	if (KIND == KIND_REDFT10) col_post2(a, buf, bout, valid, tid, nthr);
	else col_unpack3(a, buf, bout, valid, tid, nthr);
}

template <int KIND, class R, int MAXT>
__global__ void __launch_bounds__(MAXT) blue_kernel(const BlueArgsT<R> a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	cx<R> *A = reinterpret_cast<cx<R> *>(lds);
	const int tid = threadIdx.x, nthr = blockDim.x;
	long long bin, bout; int valid;
	col_base(a, blockIdx.x, bin, bout, valid);
	if (KIND == KIND_REDFT10) col_load2(a, A, bin, valid, tid, nthr);
	else col_pre3(a, A, bin, valid, tid, nthr);
	__syncthreads();
	blue_chirp_in(a, A, tid, nthr);
	__syncthreads();
	for (int s = 0; s < a.fftM.ns; s++) {
		fft_stage(A, a.M, a.fftM.st[s], a.B, a.divB, a.WM, tid, nthr);
		__syncthreads();
	}
	blue_mul(a, A, tid, nthr);
	__syncthreads();
	for (int s = a.fftM.ns - 1; s >= 0; s--) {
		fft_stage_inv(A, a.M, a.fftM.st[s], a.B, a.divB, a.WM, tid, nthr);
		__syncthreads();
	}
	blue_chirp_out(a, A, tid, nthr);
	__syncthreads();
	if (KIND == KIND_REDFT10) col_post2(a, A, bout, valid, tid, nthr);
	else col_unpack3(a, A, bout, valid, tid, nthr);
}

template <int N, int KIND, class R>
__global__ void __launch_bounds__(256) tiny_kernel(const TinyArgsT<R> a)
{
	const long long line = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (line < a.nlines) tiny_line<N, KIND>(a, line);
}

template <int N, int KIND, class R>
__global__ void __launch_bounds__(TINY_CHUNK) tiny_row_kernel(const TinyArgsT<R> a)
{
	__shared__ R lds[TINY_CHUNK * tiny_pitch<N>()];
	long long bin, bout; int cnt;
	tiny_row_base(a, blockIdx.x, bin, bout, cnt);
	tiny_row_load<N>(a, lds, bin, cnt, threadIdx.x, TINY_CHUNK);
	__syncthreads();
	tiny_row_compute<N, KIND>(a, lds, cnt, threadIdx.x);
	__syncthreads();
	tiny_row_store<N>(a, lds, bout, cnt, threadIdx.x, TINY_CHUNK);
}

template <class R>
__global__ void __launch_bounds__(256) dense_kernel(const DenseArgsT<R> a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	R *x = reinterpret_cast<R *>(lds);
	long long bin, bout;
	dense_base(a, blockIdx.x, bin, bout);
	dense_load(a, x, bin, threadIdx.x, blockDim.x);
	__syncthreads();
	dense_compute(a, x, bout, threadIdx.x, blockDim.x);
}

// lines longer than LDS holds (DenseArgsT::stage): copy every line into the plan's staging array, then sum from there
template <class R>
__global__ void __launch_bounds__(256) dense_stage_kernel(const DenseArgsT<R> a)
{
	long long bin, bout;
	dense_base(a, blockIdx.x, bin, bout);
	dense_load(a, a.stage + (size_t)blockIdx.x * a.N, bin, threadIdx.x, blockDim.x);
}
template <class R>
__global__ void __launch_bounds__(256) dense_staged_kernel(const DenseArgsT<R> a)
{
	long long bin, bout;
	dense_base(a, blockIdx.x, bin, bout);
	dense_compute(a, a.stage + (size_t)blockIdx.x * a.N, bout, threadIdx.x, blockDim.x);
}

// ---------------------------------------------------------------------------------------------
__global__ void zigzag_kernel(uint32_t *lin, uint32_t w, uint32_t h, uint64_t first, uint64_t count)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x)
		lin[i] = zigzag_lin(w, h, first + i);
}
__global__ void zigzag_ids_kernel(uint32_t *ids, uint32_t w, uint32_t h, uint64_t step, uint64_t count)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t p = zigzag_lin(w, h, i);
		ids[p] = !step ? (uint32_t)i : p ? (uint32_t)(i / step) : 0xffffffffu;   // step 0: the owner index itself; frame ids: the DC pixel never matches (scan.c:445 clears it)
	}
}
__global__ void scatter_kernel(float *recon, const float *coeffs, const uint32_t *lin, uint64_t count, int ch)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t p = lin[i];
		if (p) for (int z = 0; z < ch; z++) recon[p * ch + z] = coeffs[p * ch + z];
	}
}
__global__ void accumulate_kernel(float *sum, const float *img, uint64_t len)
{
	const uint64_t n4 = len / 4;
	float4 *s4 = reinterpret_cast<float4 *>(sum);
	const float4 *i4 = reinterpret_cast<const float4 *>(img);
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	for (uint64_t i = t; i < n4; i += stride) {
		float4 a = s4[i]; const float4 b = i4[i];
		a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; s4[i] = a;
	}
	for (uint64_t i = n4 * 4 + t; i < len; i += stride) sum[i] += img[i];
}
__global__ void accumulate_scalar_kernel(float *sum, const float *img, uint64_t len)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) sum[i] += img[i];
}
__global__ void broadcast_kernel(float *sum, const float *c, uint64_t npix, int ch)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < npix * ch; i += (uint64_t)gridDim.x * blockDim.x) sum[i] = c[i % ch];
}
__global__ void u8_to_f32_kernel(float *d, const uint8_t *s, uint64_t len)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) d[i] = (float)s[i];
}
__global__ void f32_to_u8_kernel(uint8_t *d, const float *s, double mul, uint64_t len)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) d[i] = quantise_u8((double)s[i] * mul);
}

struct Region3 { int n[3]; long long sd[3], ss[3]; };
__global__ void region_u8_to_f32_kernel(float *d, const uint8_t *s, Region3 r)
{
	const uint64_t total = (uint64_t)r.n[0] * r.n[1] * r.n[2];
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t x = i % r.n[2], y = (i / r.n[2]) % r.n[1], z = i / ((uint64_t)r.n[2] * r.n[1]);
		d[z * r.sd[0] + y * r.sd[1] + x * r.sd[2]] = (float)s[z * r.ss[0] + y * r.ss[1] + x * r.ss[2]];
	}
}
__global__ void region_f32_to_u8_kernel(uint8_t *d, const float *s, double mul, Region3 r)
{
	const uint64_t total = (uint64_t)r.n[0] * r.n[1] * r.n[2];
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t x = i % r.n[2], y = (i / r.n[2]) % r.n[1], z = i / ((uint64_t)r.n[2] * r.n[1]);
		d[z * r.sd[0] + y * r.sd[1] + x * r.sd[2]] = quantise_u8((double)s[z * r.ss[0] + y * r.ss[1] + x * r.ss[2]] * mul);
	}
}

// ---------------------------------------------------------------------------------------------
void *be_alloc(size_t bytes) { void *p = nullptr; if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr; return p; }
void be_free(void *p) { if (p) (void)hipFree(p); }
int be_upload(void *dst, const void *src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }
int be_download(void *dst, const void *src, size_t bytes, void *stream) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return 0; }
void *be_event_create() { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; return (void *)e; }
void be_event_destroy(void *e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
int be_event_record(void *e, void *stream) { HIPCHK(hipEventRecord((hipEvent_t)e, (hipStream_t)stream)); return 0; }
int be_event_synchronize(void *e) { HIPCHK(hipEventSynchronize((hipEvent_t)e)); return 0; }
int be_event_elapsed_ms(void *a, void *b, float *ms) { HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b)); return 0; }
void *be_stream_create() { hipStream_t s = nullptr; if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr; return (void *)s; }
void be_stream_destroy(void *s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
int be_stream_synchronize(void *s) { HIPCHK(hipStreamSynchronize((hipStream_t)s)); return 0; }
void *be_order_event_create() { hipEvent_t e = nullptr; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr; return (void *)e; }
int be_stream_wait_event(void *stream, void *e) { HIPCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)e, 0)); return 0; }
size_t be_max_lds() { return 160 * 1024; }
const char *be_name() { return "hip-gfx950"; }


template <class K, class A>
static int launch_lds_kernel(K kernel, const A &a, const LaunchGeom &g, void *stream)
{
	if (int rc = allow_lds(kernel, g.lds_bytes)) return rc;
	hipLaunchKernelGGL(kernel, dim3(g.nwg), dim3(g.nthr), g.lds_bytes, (hipStream_t)stream, a);
	HIPCHK(hipGetLastError());
	return 0;
}

template <class R>
static int launch_row(const PassArgsT<R> &a, const LaunchGeom &g, void *stream)
{
	if (a.kind == KIND_REDFT10) return g.nthr > 512 ? launch_lds_kernel(row_kernel<KIND_REDFT10, R, 1024>, a, g, stream) : launch_lds_kernel(row_kernel<KIND_REDFT10, R, 512>, a, g, stream);
	return g.nthr > 512 ? launch_lds_kernel(row_kernel<KIND_REDFT01, R, 1024>, a, g, stream) : launch_lds_kernel(row_kernel<KIND_REDFT01, R, 512>, a, g, stream);
}

template <class R>
static int launch_col(const PassArgsT<R> &a, const LaunchGeom &g, void *stream)
{
	if (a.kind == KIND_REDFT10) return g.nthr > 512 ? launch_lds_kernel(col_kernel<KIND_REDFT10, R, 1024>, a, g, stream) : launch_lds_kernel(col_kernel<KIND_REDFT10, R, 512>, a, g, stream);
	return g.nthr > 512 ? launch_lds_kernel(col_kernel<KIND_REDFT01, R, 1024>, a, g, stream) : launch_lds_kernel(col_kernel<KIND_REDFT01, R, 512>, a, g, stream);
}

template <class R>
static int launch_dense(const DenseArgsT<R> &a, const LaunchGeom &g, void *stream)
{
	if (a.stage) {
		hipLaunchKernelGGL((dense_stage_kernel<R>), dim3(g.nwg), dim3(g.nthr), 0, (hipStream_t)stream, a);
		hipLaunchKernelGGL((dense_staged_kernel<R>), dim3(g.nwg), dim3(g.nthr), 0, (hipStream_t)stream, a);
		HIPCHK(hipGetLastError());
		return 0;
	}
	if (int rc = allow_lds(dense_kernel<R>, g.lds_bytes)) return rc;
	hipLaunchKernelGGL((dense_kernel<R>), dim3(g.nwg), dim3(g.nthr), g.lds_bytes, (hipStream_t)stream, a);
	HIPCHK(hipGetLastError());
	return 0;
}

template <class R>
static int launch_blue(const BlueArgsT<R> &a, const LaunchGeom &g, void *stream)
{
	if (a.kind == KIND_REDFT10) return g.nthr > 512 ? launch_lds_kernel(blue_kernel<KIND_REDFT10, R, 1024>, a, g, stream) : launch_lds_kernel(blue_kernel<KIND_REDFT10, R, 512>, a, g, stream);
	return g.nthr > 512 ? launch_lds_kernel(blue_kernel<KIND_REDFT01, R, 1024>, a, g, stream) : launch_lds_kernel(blue_kernel<KIND_REDFT01, R, 512>, a, g, stream);
}
int be_launch_blue(const BlueArgs &a, const LaunchGeom &g, void *stream) { return launch_blue(a, g, stream); }
int be_launch_blue(const BlueArgsD &a, const LaunchGeom &g, void *stream) { return launch_blue(a, g, stream); }

template <int N, class R>
static int launch_tiny_n(const TinyArgsT<R> &a, void *stream)
{
	if (a.packed) {
		long long rest = 1;
		for (int d = 1; d < a.nd; d++) rest *= a.bn[d];
		const unsigned grid = (unsigned)(rest * a.chunk_div.d);
		if (a.kind == KIND_REDFT10) hipLaunchKernelGGL((tiny_row_kernel<N, KIND_REDFT10, R>), dim3(grid), dim3(TINY_CHUNK), 0, (hipStream_t)stream, a);
		else hipLaunchKernelGGL((tiny_row_kernel<N, KIND_REDFT01, R>), dim3(grid), dim3(TINY_CHUNK), 0, (hipStream_t)stream, a);
		HIPCHK(hipGetLastError());
		return 0;
	}
	const unsigned grid = (unsigned)((a.nlines + 255) / 256);
	if (a.kind == KIND_REDFT10) hipLaunchKernelGGL((tiny_kernel<N, KIND_REDFT10, R>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
	else hipLaunchKernelGGL((tiny_kernel<N, KIND_REDFT01, R>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
	HIPCHK(hipGetLastError());
	return 0;
}
template <class R>
static int launch_tiny(const TinyArgsT<R> &a, void *stream)
{
	switch (a.N) {
#define DSP_TINY_CASE(n) case n: return launch_tiny_n<n>(a, stream);
	DSP_TINY_CASE(1) DSP_TINY_CASE(2) DSP_TINY_CASE(3) DSP_TINY_CASE(4) DSP_TINY_CASE(5) DSP_TINY_CASE(6) DSP_TINY_CASE(7) DSP_TINY_CASE(8)
	DSP_TINY_CASE(9) DSP_TINY_CASE(10) DSP_TINY_CASE(11) DSP_TINY_CASE(12) DSP_TINY_CASE(13) DSP_TINY_CASE(14) DSP_TINY_CASE(15) DSP_TINY_CASE(16)
	DSP_TINY_CASE(17) DSP_TINY_CASE(18) DSP_TINY_CASE(19) DSP_TINY_CASE(20) DSP_TINY_CASE(21) DSP_TINY_CASE(22) DSP_TINY_CASE(23) DSP_TINY_CASE(24)
	DSP_TINY_CASE(25) DSP_TINY_CASE(26) DSP_TINY_CASE(27) DSP_TINY_CASE(28) DSP_TINY_CASE(29) DSP_TINY_CASE(30) DSP_TINY_CASE(31) DSP_TINY_CASE(32)
#undef DSP_TINY_CASE
	default: return -1;
	}
}
int be_launch_tiny(const TinyArgs &a, void *stream) { return launch_tiny(a, stream); }
int be_launch_tiny(const TinyArgsD &a, void *stream) { return launch_tiny(a, stream); }

int be_launch_row(const PassArgs &a, const LaunchGeom &g, void *stream) { return launch_row(a, g, stream); }
int be_launch_col(const PassArgs &a, const LaunchGeom &g, void *stream) { return launch_col(a, g, stream); }
int be_launch_dense(const DenseArgs &a, const LaunchGeom &g, void *stream) { return launch_dense(a, g, stream); }
int be_launch_row(const PassArgsD &a, const LaunchGeom &g, void *stream) { return launch_row(a, g, stream); }
int be_launch_col(const PassArgsD &a, const LaunchGeom &g, void *stream) { return launch_col(a, g, stream); }
int be_launch_dense(const DenseArgsD &a, const LaunchGeom &g, void *stream) { return launch_dense(a, g, stream); }

#include "spec_registry.inc"

__global__ void motion_filter_span_kernel(float *c, MotionFilter p, uint64_t span, unsigned long long *coded)
{
	unsigned long long mine = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < span; i += (uint64_t)gridDim.x * blockDim.x)
		c[i] = motion_filter_elem(p, (uint32_t)i, c[i], mine);
	if (coded) {
		for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
		if ((threadIdx.x & 63) == 0 && mine) atomicAdd(coded, mine);
	}
}
int be_motion_filter(float *buf, const MotionFilter &filt, uint64_t span, unsigned long long *coded, void *stream)
{
	if (!span) return 0;
	uint64_t b = (span + 255) / 256;
	hipLaunchKernelGGL(motion_filter_span_kernel, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(256), 0, (hipStream_t)stream, buf, filt, span, coded);
	HIPCHK(hipGetLastError());
	return 0;
}

__global__ void zoomx_table_kernel(float *tab, int M, int cw, int nsrc, double theta, double scale)
{
	zoomx_table_item(tab, M, cw, nsrc, theta, scale, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}
int be_zoomx_tables(float *tab, int M, int cw, int nsrc, double theta, double scale, void *stream)
{
	const int items = nsrc * (M / 4 + 1);
	hipLaunchKernelGGL(zoomx_table_kernel, dim3((items + 127) / 128), dim3(128), 0, (hipStream_t)stream, tab, M, cw, nsrc, theta, scale);
	HIPCHK(hipGetLastError());
	return 0;
}

__global__ void czt_tables_kernel(cf *atab, cf *etab, cf *htab, int nc, int nout, int P, double omega, double phi, double scale)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (atab && i < nc) atab[i] = czt_a_entry(i, omega, phi);
	if (etab && i < nout) etab[i] = czt_e_entry(i, omega, scale);
	if (htab && i < P) htab[i] = czt_h_entry(i, P, nc, nout, omega);
}
int be_czt_tables(cf *atab, cf *etab, cf *htab, int nc, int nout, int P, double omega, double phi, double scale, void *stream)
{
	const int n = P > nout ? (P > nc ? P : nc) : (nout > nc ? nout : nc);
	hipLaunchKernelGGL(czt_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, atab, etab, htab, nc, nout, P, omega, phi, scale);
	HIPCHK(hipGetLastError());
	return 0;
}
// out[c][r] = in[r][c] for r < rows, c < cols (pitches in floats): 32 x 32 tiles through LDS
__global__ void transpose_kernel(float *out, long long out_pitch, const float *in, long long in_pitch, int rows, int cols)
{
	__shared__ float tile[32][33];
	const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
	for (int i = ty; i < 32; i += 8) if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(long long)(r0 + i) * in_pitch + c0 + tx];
	__syncthreads();
	for (int i = ty; i < 32; i += 8) if (c0 + i < cols && r0 + tx < rows) out[(long long)(c0 + i) * out_pitch + r0 + tx] = tile[tx][i];
}
int be_transpose(float *out, long long out_pitch, const float *in, long long in_pitch, int rows, int cols, void *stream)
{
	hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, out, out_pitch, in, in_pitch, rows, cols);
	HIPCHK(hipGetLastError());
	return 0;
}

static inline int ew_grid(uint64_t n) { uint64_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

int be_scan_zigzag(uint32_t *lin, uint32_t w, uint32_t h, uint64_t first, uint64_t count, void *stream)
{
	if (!count) return 0;
	hipLaunchKernelGGL(zigzag_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, lin, w, h, first, count);
	HIPCHK(hipGetLastError());
	return 0;
}
int be_scan_zigzag_frame_ids(uint32_t *ids, uint32_t w, uint32_t h, uint64_t step, void *stream)
{
	const uint64_t count = (uint64_t)w * h;
	hipLaunchKernelGGL(zigzag_ids_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, ids, w, h, step, count);
	HIPCHK(hipGetLastError());
	return 0;
}
int be_scan_scatter(float *recon, const float *coeffs, const uint32_t *lin, uint64_t count, uint64_t npixels, int ch, void *stream)
{
	HIPCHK(hipMemsetAsync(recon, 0, sizeof(float) * npixels * ch, (hipStream_t)stream));
	if (!count) return 0;
	hipLaunchKernelGGL(scatter_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, recon, coeffs, lin, count, ch);
	HIPCHK(hipGetLastError());
	return 0;
}
int be_accumulate(float *sum, const float *img, uint64_t len, void *stream)
{
	if (!len) return 0;
	if ((((uintptr_t)sum | (uintptr_t)img) & 15) == 0)
		hipLaunchKernelGGL(accumulate_kernel, dim3(ew_grid(len / 4 + 1)), dim3(256), 0, (hipStream_t)stream, sum, img, len);
	else
		hipLaunchKernelGGL(accumulate_scalar_kernel, dim3(ew_grid(len)), dim3(256), 0, (hipStream_t)stream, sum, img, len);
	HIPCHK(hipGetLastError());
	return 0;
}
int be_broadcast_dc(float *sum, const float *c, uint64_t npix, int ch, void *stream)
{
	if (!npix) return 0;
	hipLaunchKernelGGL(broadcast_kernel, dim3(ew_grid(npix * ch)), dim3(256), 0, (hipStream_t)stream, sum, c, npix, ch);
	HIPCHK(hipGetLastError());
	return 0;
}
int be_u8_to_f32(float *d, const uint8_t *s, uint64_t len, void *stream)
{
	if (!len) return 0;
	hipLaunchKernelGGL(u8_to_f32_kernel, dim3(ew_grid(len)), dim3(256), 0, (hipStream_t)stream, d, s, len);
	HIPCHK(hipGetLastError());
	return 0;
}
int be_zero(void *p, size_t bytes, void *stream) { HIPCHK(hipMemsetAsync(p, 0, bytes, (hipStream_t)stream)); return 0; }
static Region3 make_region(const int n[3], const long long sd[3], const long long ss[3])
{
	Region3 r;
	for (int a = 0; a < 3; a++) { r.n[a] = n[a]; r.sd[a] = sd[a]; r.ss[a] = ss[a]; }
	return r;
}
int be_region_u8_to_f32(float *dst, const uint8_t *src, const int n[3], const long long sdst[3], const long long ssrc[3], void *stream)
{
	const uint64_t total = (uint64_t)n[0] * n[1] * n[2];
	if (!total) return 0;
	hipLaunchKernelGGL(region_u8_to_f32_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dst, src, make_region(n, sdst, ssrc));
	HIPCHK(hipGetLastError());
	return 0;
}
int be_region_f32_to_u8(uint8_t *dst, const float *src, double mul, const int n[3], const long long sdst[3], const long long ssrc[3], void *stream)
{
	const uint64_t total = (uint64_t)n[0] * n[1] * n[2];
	if (!total) return 0;
	hipLaunchKernelGGL(region_f32_to_u8_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dst, src, mul, make_region(n, sdst, ssrc));
	HIPCHK(hipGetLastError());
	return 0;
}
int be_f32_to_u8(uint8_t *d, const float *s, double mul, uint64_t len, void *stream)
{
	if (!len) return 0;
	hipLaunchKernelGGL(f32_to_u8_kernel, dim3(ew_grid(len)), dim3(256), 0, (hipStream_t)stream, d, s, mul, len);
	HIPCHK(hipGetLastError());
	return 0;
}

}  // namespace dspfft
