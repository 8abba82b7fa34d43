// zoom_gemm.hip -- zoom's dense separable basis product (zoom/zoom.c:36-68 basis, :361-375 product)
// on the f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact f32 multiply-accumulate at the
// vector-FP32 rate, 157 TFLOP/s peak).  This is the one place on the hot path that is a true dense
// GEMM: at BASELINE config 3 it is 310.6 GFLOP against ~0.45 GB of operands.
//
//   per channel z:   Tt = XB (vw x cw) . C_z^T (cw x ch)          -> Tt  (vw x ch)
//                    out_z = YB (vh x ch) . Tt^T (ch x vw) / (w h) -> out (vh x vw), interleaved store
// where XB/YB are the reference's bases with the halved DC term folded in as column 0 = 1/2
// (zoom.c:364 `tmp = C[row][0]/2`, :369 `s = tmp[0]/2`).  Both products are "NT" GEMMs (both
// operands have the summed index contiguous), so one kernel serves both.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/dspfft.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128;

// C[m*ldc + n*cs] = alpha * sum_k A[m*lda + k] * B[n*ldb + k];  batch b offsets: sa, sb, sc
// 128 x 128 x BK block tile, 256 threads = 2 x 2 waves of 64 x 64, each 2 x 2 MFMA 32x32x2 tiles;
// global -> registers -> LDS double buffer (the next K-tile's loads are in flight during the MFMAs).
// LDS tiles are [row][k] with k contiguous: a thread's float4 along K goes in with ONE ds_write_b128 (no transposing
// scalar writes) and every lane takes its MFMA operands for four k-steps with ONE ds_read_b128.  The two k-slots of
// an MFMA need not be adjacent in memory, only the same for A and B: lane half lk owns k = 4 lk .. 4 lk + 3 of each
// group of eight.  Row pitch BK + 4 floats keeps the b128 accesses of a wave spread over the banks.
template <int BK>
__global__ void __launch_bounds__(256) gemm_nt_f32_mfma(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                        int M, int N, int K, long long lda, long long ldb, long long ldc, int cs,
                                                        long long sa, long long sb, long long sc, float alpha)
{
	constexpr int PITCH = BK + 4;
	__shared__ __attribute__((aligned(16))) float As[2][BM][PITCH];
	__shared__ __attribute__((aligned(16))) float Bs[2][BN][PITCH];
	constexpr int NV = BK / 8;                               // float4 per thread per operand per K-tile
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1;                 // 2 x 2 waves, each 64 x 64
	const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;
	A += (long long)blockIdx.z * sa; B += (long long)blockIdx.z * sb; C += (long long)blockIdx.z * sc;

	// staging: thread -> (row = tid / 2, k4 = (tid % 2) * 4 + 8 v): float4 along K
	const int srow = tid >> 1, sk = (tid & 1) * 4;
	const bool a_ok = bm + srow < M, b_ok = bn + srow < N;
	const float *ap = A + (long long)(bm + srow) * lda + sk;
	const float *bp = B + (long long)(bn + srow) * ldb + sk;
	const bool vec = ((lda | ldb | sa | sb) & 3) == 0 && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0;

	auto fetch = [&](const float *p, bool ok, int k0) -> float4 {
		float4 v; v.x = v.y = v.z = v.w = 0.f;
		if (!ok) return v;
		if (vec && k0 + sk + 3 < K) return *reinterpret_cast<const float4 *>(p + k0);
		if (k0 + sk + 0 < K) v.x = p[k0 + 0];
		if (k0 + sk + 1 < K) v.y = p[k0 + 1];
		if (k0 + sk + 2 < K) v.z = p[k0 + 2];
		if (k0 + sk + 3 < K) v.w = p[k0 + 3];
		return v;
	};
	float4 ra[NV], rb[NV];
	auto fetch_tile = [&](int k0) {
#pragma unroll
		for (int v = 0; v < NV; v++) { ra[v] = fetch(ap, a_ok, k0 + 8 * v); rb[v] = fetch(bp, b_ok, k0 + 8 * v); }
	};
	auto stash = [&](int buf) {
#pragma unroll
		for (int v = 0; v < NV; v++) {
			*reinterpret_cast<float4 *>(&As[buf][srow][sk + 8 * v]) = ra[v];
			*reinterpret_cast<float4 *>(&Bs[buf][srow][sk + 8 * v]) = rb[v];
		}
	};

	f32x16 acc[2][2];
	for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

	fetch_tile(0);
	stash(0);
	__syncthreads();
	const int nk = (K + BK - 1) / BK;
	const int li = lane & 31, lk = lane >> 5;
	for (int kt = 0; kt < nk; kt++) {
		const int cur = kt & 1;
		if (kt + 1 < nk) fetch_tile((kt + 1) * BK);
#pragma unroll
		for (int g = 0; g < BK / 8; g++) {
			const float4 a0 = *reinterpret_cast<const float4 *>(&As[cur][wm * 64 + li][8 * g + 4 * lk]);
			const float4 a1 = *reinterpret_cast<const float4 *>(&As[cur][wm * 64 + 32 + li][8 * g + 4 * lk]);
			const float4 b0 = *reinterpret_cast<const float4 *>(&Bs[cur][wn * 64 + li][8 * g + 4 * lk]);
			const float4 b1 = *reinterpret_cast<const float4 *>(&Bs[cur][wn * 64 + 32 + li][8 * g + 4 * lk]);
			const float a0v[4] = {a0.x, a0.y, a0.z, a0.w}, a1v[4] = {a1.x, a1.y, a1.z, a1.w};
			const float b0v[4] = {b0.x, b0.y, b0.z, b0.w}, b1v[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
			for (int s = 0; s < 4; s++) {
				acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[s], b0v[s], acc[0][0], 0, 0, 0);
				acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[s], b1v[s], acc[0][1], 0, 0, 0);
				acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[s], b0v[s], acc[1][0], 0, 0, 0);
				acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[s], b1v[s], acc[1][1], 0, 0, 0);
			}
		}
		if (kt + 1 < nk) stash(cur ^ 1);
		__syncthreads();
	}
	// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
	for (int i = 0; i < 2; i++)
		for (int j = 0; j < 2; j++) {
			const int n = bn + wn * 64 + j * 32 + li;
			if (n >= N) continue;
			for (int r = 0; r < 16; r++) {
				const int m = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
				if (m < M) C[(long long)m * ldc + (long long)n * cs] = alpha * acc[i][j][r];
			}
		}
}

// zoom/zoom.c:36-68 with column 0 = 1/2 (the halved DC term) and columns 1.. = the reference's basis
__global__ void zoom_basis_kernel(float *basis, int type, double scale_num, double scale_den, double offset, size_t nvectors, size_t len, size_t nc)
{
	const size_t total = nvectors * nc;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const size_t b = i / nc, n = i - b * nc;
		if (n == 0) { basis[i] = 0.5f; continue; }
		double k, N;
		if (type == 2) { k = b + offset; N = len * scale_num / scale_den; }                                   /* native  (zoom.c:50-53) */
		else if (type == 0) { k = (b + offset) * scale_den / scale_num; N = (double)len; }                    /* interpolated (:54-57) */
		else { k = (b + offset) * (len - 1) * scale_den / (len * scale_num - scale_den); N = (double)len; }   /* centered (:58-61) */
		basis[i] = (float)cos(M_PI * (k + 0.5) * (double)n / N);
	}
}

__global__ void deinterleave3_kernel(float *planes, const float *img, size_t npix)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix * 3; i += (size_t)gridDim.x * blockDim.x) {
		const size_t p = i / 3, z = i - p * 3;
		planes[z * npix + p] = img[i];
	}
}

// ---- applybasis (applybasis/applybasis.c:77-140): basis matrices F[k][n] = f(k + offset, n, N, ortho) ----
__device__ void ab_basis(int func, long long k, long long n, unsigned long long N, int ortho, double &re, double &im)
{
	const double r2 = 1.41421356237309504880, pi = 3.14159265358979323846;
	double c; im = 0.0;
	switch (func) {
	case 0: { double a = (-2.0 * pi * (double)(k * n)) / (double)N; re = cos(a); im = sin(a); return; }
	case 1: { double a = (2.0 * pi * (double)(k * n)) / (double)N; re = cos(a); im = sin(a); return; }
	case 2: c = (n && N - 1 - n) ? cos((pi * (double)(k * n)) / (double)(N - 1)) : (n ? ((k & 1) ? -1.0 : 1.0) : 1.0) / 2; if (ortho) c *= r2; break;
	case 3: c = cos((pi * (double)(k * (2 * n + 1))) / (double)(2 * N)); if (ortho) c *= (k ? r2 : 1); break;
	case 4: c = n ? cos((pi * (double)(n * (2 * k + 1))) / (double)(2 * N)) : 0.5; if (ortho) c *= n ? r2 : 2; break;
	case 5: c = cos((pi * (double)((2 * k + 1) * (2 * n + 1))) / (double)(4 * N)); if (ortho) c *= r2; break;
	case 6: c = sin((pi * (double)((k + 1) * (n + 1))) / (double)(N + 1)); if (ortho) c *= r2; break;
	case 7: c = sin((pi * (double)((k + 1) * (2 * n + 1))) / (double)(2 * N)); if (ortho) c *= (N - 1 - k) ? r2 : 1; break;
	case 8: c = (N - 1 - n) ? sin((pi * (double)((2 * k + 1) * (n + 1))) / (double)(2 * N)) : ((k & 1) ? -1.0 : 1.0) / 2; if (ortho) c *= (N - 1 - n) ? r2 : 2; break;
	case 9: c = sin((pi * (double)((2 * k + 1) * (2 * n + 1))) / (double)(4 * N)); if (ortho) c *= r2; break;
	case 10: {
		unsigned long long L = (unsigned long long)log2((double)N), nn = (unsigned long long)n, kk = (unsigned long long)k;
		unsigned long long sig = (nn & (kk >> (L - 1))) & 1ULL;
		for (L--, nn >>= 1; L; L--, nn >>= 1) sig += (nn & ((kk >> (L - 1)) + (kk >> L))) & 1ULL;
		c = (sig & 1) ? -1.0 : 1.0; break;
	}
	default: c = r2 * cos(2 * pi * (double)n * (double)k / (double)N - pi / 4); break;
	}
	re = c;
}
__global__ void ab_basis_kernel(float *re, float *im, int func, int ortho, long long terms, long long offset, unsigned long long N)
{
	const size_t total = (size_t)terms * N;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const long long k = (long long)(i / N), n = (long long)(i - (size_t)k * N);
		double r, m; ab_basis(func, k + offset, n, N, ortho, r, m);
		re[i] = (float)r; if (im) im[i] = (float)m;
	}
}
// out[kh][kw][nh][nw][j] (re, im)  <-  P[c][kh][nh][kw*Nw+nw] products (see dspfft_applybasis_partsums)
__global__ void ab_combine_kernel(float *out, const float *P, int cplx, int Kh, int Kw, int Nh, int Nw)
{
	const size_t per = (size_t)Kh * Nh * Kw * Nw, total = per * 3;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const size_t j = i / per, r = i - j * per;               // channel-major temp
		const size_t kh = r / ((size_t)Nh * Kw * Nw), r2 = r - kh * ((size_t)Nh * Kw * Nw);
		const size_t nh = r2 / ((size_t)Kw * Nw), r3 = r2 - nh * ((size_t)Kw * Nw);
		const size_t kw = r3 / Nw, nw = r3 - kw * Nw;
		const size_t o = ((((kh * Kw + kw) * Nh + nh) * Nw + nw) * 3 + j) * 2;
		if (cplx) {
			// P blocks: 0 = Fh_re.T_re, 1 = Fh_im.T_im, 2 = Fh_re.T_im, 3 = Fh_im.T_re   (each 3*per floats)
			out[o] = P[i] - P[3 * per + i];
			out[o + 1] = P[2 * 3 * per + i] + P[3 * 3 * per + i];
		} else { out[o] = P[i]; out[o + 1] = 0.f; }
	}
}

thread_local char g_zerr[256] = "";

}  // namespace

extern "C" const char *dspfft_zoom_last_error(void) { return g_zerr; }

extern "C" size_t dspfft_zoom_ncomponents(double scale_num, double scale_den, size_t len)
{
	if (len * scale_num / scale_den < 1) { scale_num = 1; scale_den = (double)len; }      /* zoom.c:37-40 */
	const double want = round(len * scale_num / scale_den);
	return want < (double)len ? (size_t)want : len;                                       /* zoom.c:41 */
}

extern "C" int dspfft_zoom_basis(float *d_basis, int type, double scale_num, double scale_den, double offset,
                                 size_t nvectors, size_t len, void *stream)
{
	if (!d_basis || type < 0 || type > 2 || !nvectors || !len) { snprintf(g_zerr, sizeof g_zerr, "bad arguments"); return -1; }
	if (len * scale_num / scale_den < 1) { scale_num = 1; scale_den = (double)len; }
	const size_t nc = dspfft_zoom_ncomponents(scale_num, scale_den, len);
	hipLaunchKernelGGL(zoom_basis_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, d_basis, type, scale_num, scale_den, offset, nvectors, len, nc);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int dspfft_gemm_nt_f32(const float *A, const float *B, float *C, int M, int N, int K,
                                  long long lda, long long ldb, long long ldc, int cs,
                                  int batch, long long sa, long long sb, long long sc, float alpha, void *stream)
{
	if (!A || !B || !C || M < 1 || N < 1 || K < 1 || batch < 1 || cs < 1) { snprintf(g_zerr, sizeof g_zerr, "bad arguments"); return -1; }
	dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
	static const int bk = getenv("DSPFFT_GEMM_BK") ? atoi(getenv("DSPFFT_GEMM_BK")) : 8;
	if (K >= 32 && bk == 32) hipLaunchKernelGGL(gemm_nt_f32_mfma<32>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, cs, sa, sb, sc, alpha);
	else if (K >= 16 && bk >= 16) hipLaunchKernelGGL(gemm_nt_f32_mfma<16>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, cs, sa, sb, sc, alpha);
	else hipLaunchKernelGGL(gemm_nt_f32_mfma<8>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, cs, sa, sb, sc, alpha);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" size_t dspfft_zoom_work_floats(int w, int h, size_t ch, int vw)
{
	return (size_t)3 * w * h + (size_t)3 * vw * ch;      // planar copy of the coefficients + Tt for 3 channels
}

extern "C" int dspfft_zoom_product(const float *d_coeffs, int w, int h, const float *d_xb, size_t cw, const float *d_yb, size_t ch,
                                   float *d_out, int vw, int vh, float *d_work, void *stream)
{
	if (!d_coeffs || !d_xb || !d_yb || !d_out || !d_work || cw < 1 || ch < 1 || cw > (size_t)w || ch > (size_t)h) { snprintf(g_zerr, sizeof g_zerr, "bad arguments"); return -1; }
	const size_t npix = (size_t)w * h;
	float *planes = d_work, *Tt = d_work + 3 * npix;
	hipLaunchKernelGGL(deinterleave3_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, planes, d_coeffs, npix);
	// Tt_z (vw x ch) = XB (vw x cw) . plane_z[:ch, :cw]^T
	int rc = dspfft_gemm_nt_f32(d_xb, planes, Tt, vw, (int)ch, (int)cw, (long long)cw, w, (long long)ch, 1, 3, 0, (long long)npix, (long long)vw * ch, 1.f, stream);
	if (rc) return rc;
	// out_z (vh x vw, interleaved) = YB (vh x ch) . Tt_z^T / (w h)
	return dspfft_gemm_nt_f32(d_yb, Tt, d_out, vh, vw, (int)ch, (long long)ch, (long long)ch, (long long)vw * 3, 3, 3, 0, (long long)vw * ch, 1, 1.f / ((float)w * (float)h), stream);
}

// ---- applybasis' forward partial sums (applybasis/applybasis.c:410-431) as batched NT GEMMs ----
extern "C" size_t dspfft_applybasis_work_floats(int w, int h, int Kw, int Kh, int Pw, int Ph, int func)
{
	const size_t Nw = (size_t)w / Pw, Nh = (size_t)h / Ph, cplx = func <= 1 ? 2 : 1;
	return (size_t)3 * w * h                       /* planar pixels */
	     + cplx * ((size_t)Kw * w + (size_t)Kh * h) /* basis matrices */
	     + cplx * 3 * (size_t)Kw * Nw * h           /* Tt */
	     + cplx * cplx * 3 * (size_t)Kh * Nh * Kw * Nw; /* products */
}

extern "C" int dspfft_applybasis_partsums(float *d_out, const float *d_pixels, int w, int h, int func, int ortho,
                                          int Kw, int Kh, int Pw, int Ph, long long offw, long long offh,
                                          float *d_work, void *stream)
{
	if (!d_out || !d_pixels || !d_work || func < 0 || func > 11 || Kw < 1 || Kh < 1 || Pw < 1 || Ph < 1 || w % Pw || h % Ph) {
		snprintf(g_zerr, sizeof g_zerr, "bad arguments (the partial-sum block must divide the image)"); return -1;
	}
	hipStream_t s = (hipStream_t)stream;
	const int Nw = w / Pw, Nh = h / Ph, cplx = func <= 1;
	const size_t npix = (size_t)w * h;
	float *planes = d_work;
	float *Fw_re = planes + 3 * npix, *Fw_im = cplx ? Fw_re + (size_t)Kw * w : nullptr;
	float *Fh_re = Fw_re + (cplx ? 2 : 1) * (size_t)Kw * w, *Fh_im = cplx ? Fh_re + (size_t)Kh * h : nullptr;
	float *Tt = Fh_re + (cplx ? 2 : 1) * (size_t)Kh * h;                       // [c?][3][Kw][Nw][h]
	const size_t tsz = (size_t)3 * Kw * Nw * h;
	float *P = Tt + (cplx ? 2 : 1) * tsz;                                       // [blocks][3][Kh][Nh][Kw*Nw]
	const size_t per = (size_t)Kh * Nh * Kw * Nw, psz = 3 * per;
	hipLaunchKernelGGL(deinterleave3_kernel, dim3(1024), dim3(256), 0, s, planes, d_pixels, npix);
	hipLaunchKernelGGL(ab_basis_kernel, dim3(512), dim3(256), 0, s, Fw_re, Fw_im, func, ortho, (long long)Kw, offw, (unsigned long long)w);
	hipLaunchKernelGGL(ab_basis_kernel, dim3(512), dim3(256), 0, s, Fh_re, Fh_im, func, ortho, (long long)Kh, offh, (unsigned long long)h);
	// step 1: Tt[j][kw][nw][y] = sum_{sw} Fw[kw][nw Pw + sw] pix_j[y][nw Pw + sw]      (batch over nw, per channel)
	for (int part = 0; part < (cplx ? 2 : 1); part++)
		for (int j = 0; j < 3; j++) {
			int rc = dspfft_gemm_nt_f32(part ? Fw_im : Fw_re, planes + j * npix, Tt + part * tsz + (size_t)j * Kw * Nw * h,
			                            Kw, h, Pw, w, w, (long long)Nw * h, 1, Nw, Pw, Pw, h, 1.f, stream);
			if (rc) return rc;
		}
	// step 2: P[j][kh][nh][kw*Nw+nw] = sum_{sh} Fh[kh][nh Ph + sh] Tt[j][kw][nw][nh Ph + sh]   (batch over nh, per channel)
	for (int hp = 0; hp < (cplx ? 2 : 1); hp++)        // Fh part
		for (int tp = 0; tp < (cplx ? 2 : 1); tp++) {  // Tt part
			// block order expected by ab_combine_kernel: 0 re.re, 1 im.im, 2 re.im, 3 im.re
			const int slot = cplx ? (hp == 0 && tp == 0 ? 0 : hp == 1 && tp == 1 ? 1 : hp == 0 ? 2 : 3) : 0;
			for (int j = 0; j < 3; j++) {
				int rc = dspfft_gemm_nt_f32(hp ? Fh_im : Fh_re, Tt + tp * tsz + (size_t)j * Kw * Nw * h, P + slot * psz + (size_t)j * per,
				                            Kh, Kw * Nw, Ph, h, h, (long long)Nh * Kw * Nw, 1, Nh, Ph, Ph, (long long)Kw * Nw, 1.f, stream);
				if (rc) return rc;
			}

		}
	hipLaunchKernelGGL(ab_combine_kernel, dim3(1024), dim3(256), 0, s, d_out, P, cplx, Kh, Kw, Nh, Nw);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}
