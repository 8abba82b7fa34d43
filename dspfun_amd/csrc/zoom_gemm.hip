// zoom_gemm.hip -- zoom's dense separable basis product (zoom/zoom.c:36-68 basis, :361-375 product)
// on the f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact f32 multiply-accumulate at the
// vector-FP32 rate, 157 TFLOP/s peak).  This is the one place on the hot path that is a true dense
// GEMM: at BASELINE config 3 it is 310.6 GFLOP against ~0.45 GB of operands.
//
//   per channel z:   Tt[3 x + z] = XB (vw x cw) . C_z^T (cw x ch)      -> Tt (3 vw x ch), the channels interleaved by row
//   once:            out = YB (vh x ch) . Tt^T (ch x 3 vw) / (w h)     -> out (vh x 3 vw) = the interleaved frame, contiguous stores
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
// TW = 2: the 128 x 128 tile above.  TW = 1: 64 x 64 (2 x 2 waves of 32 x 32, one MFMA tile each) for products too small to give every CU a
// 128 x 128 tile (applybasis' partial sums of a 512 x 512 image: 48 workgroups for 256 CUs); half of the threads stage A, the other half B.
template <int BK, int TW = 2>
__global__ void __launch_bounds__(256) gemm_nt_f32_mfma(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                        int M, int N, int K, long long lda, long long ldb, long long ldc, int cs,
                                                        long long sa, long long sb, long long sc, float alpha,
                                                        int nb1 = 0, long long sa2 = 0, long long sb2 = 0, long long sc2 = 0)
{
	constexpr int PITCH = BK + 4, BM = 64 * TW, BN = 64 * TW, WT = 32 * TW;      // WT: a wave's square of the tile
	__shared__ __attribute__((aligned(16))) float As[2][BM][PITCH];
	__shared__ __attribute__((aligned(16))) float Bs[2][BN][PITCH];
	constexpr int NV = BK / 8;                               // float4 per thread per operand per K-tile
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1;                 // 2 x 2 waves, each WT x WT
	const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;
	{	// blockIdx.z = b1 + nb1 * b2: two batch levels (e.g. partial-sum block and colour channel), nb1 = 0: one level
		const int b2 = nb1 ? (int)blockIdx.z / nb1 : 0, b1 = (int)blockIdx.z - b2 * nb1;
		A += (long long)b1 * sa + (long long)b2 * sa2; B += (long long)b1 * sb + (long long)b2 * sb2; C += (long long)b1 * sc + (long long)b2 * sc2;
	}

	// staging: thread -> (row = tid / 2, k4 = (tid % 2) * 4 + 8 v): float4 along K
	const int srow = (tid >> 1) & (BM - 1), sk = (tid & 1) * 4;
	const bool does_a = TW == 2 || tid < 128, does_b = TW == 2 || tid >= 128;
	const bool a_ok = does_a && bm + srow < M, b_ok = does_b && bn + srow < N;
	const float *ap = A + (long long)(bm + srow) * lda + sk;
	const float *bp = B + (long long)(bn + srow) * ldb + sk;
	const bool vec = ((lda | ldb | sa | sb | sa2 | sb2) & 3) == 0 && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0;

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
			if (does_a) *reinterpret_cast<float4 *>(&As[buf][srow][sk + 8 * v]) = ra[v];
			if (does_b) *reinterpret_cast<float4 *>(&Bs[buf][srow][sk + 8 * v]) = rb[v];
		}
	};

	f32x16 acc[TW][TW];
	for (int i = 0; i < TW; i++) for (int j = 0; j < TW; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

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
			float av[TW][4], bv[TW][4];
#pragma unroll
			for (int i = 0; i < TW; i++) {
				const float4 a = *reinterpret_cast<const float4 *>(&As[cur][wm * WT + 32 * i + li][8 * g + 4 * lk]);
				const float4 b = *reinterpret_cast<const float4 *>(&Bs[cur][wn * WT + 32 * i + li][8 * g + 4 * lk]);
				av[i][0] = a.x; av[i][1] = a.y; av[i][2] = a.z; av[i][3] = a.w;
				bv[i][0] = b.x; bv[i][1] = b.y; bv[i][2] = b.z; bv[i][3] = b.w;
			}
#pragma unroll
			for (int s = 0; s < 4; s++)
#pragma unroll
				for (int i = 0; i < TW; i++)
#pragma unroll
					for (int j = 0; j < TW; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
		}
		if (kt + 1 < nk) stash(cur ^ 1);
		__syncthreads();
	}
	// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
	for (int i = 0; i < TW; i++)
		for (int j = 0; j < TW; j++) {
			const int n = bn + wn * WT + j * 32 + li;
			if (n >= N) continue;
			for (int r = 0; r < 16; r++) {
				const int m = bm + wm * WT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
				if (m < M) C[(long long)m * ldc + (long long)n * cs] = alpha * acc[i][j][r];
			}
		}
}

// ---- round 4: the same product with its operands staged by LDS-DMA (global_load_lds_dwordx4) in K-tiles of 32 ----
// 256 threads = 2 x 2 waves, two LDS buffers.  The kernel above stages through registers one 8-deep K-tile ahead: 16 MFMAs (1024 cycles per
// wave) between barriers, less than a global load's latency, so every K-tile ends waiting for the next one's operands (MfmaUtil 0.58-0.71).
// Here a K-tile is 64 MFMAs per wave (4096 cycles): the DMA of tile t + 1 is issued before the MFMAs of tile t and has landed when they end;
// no staging registers, no ds_write pass.  What it took to get from there (119 TF on 8192 x 8192 x 4096) to 133 TF = MfmaUtil 0.85
// (profiles/r04_gemm.txt): the compiler must SEE that the DMA's LDS writes and the ds_reads never meet (`ktile`'s __restrict__ buffers:
// without that, s_waitcnt vmcnt(0) sits between the DMA and the first ds_read and nothing is prefetched), and the DMA's addressing must
// not cost vector instructions (`stage`): every VALU instruction a wave issues between its MFMAs is a slot the matrix pipe idles.
// LDS image of an operand tile (lane-linear per DMA instruction: its destination is base + lane x 16 bytes): [32-row block][k-group of 8]
// [k-half lk][row li] x 16 bytes, i.e. DMA lane lk * 32 + li fetches the four floats k0 + 8 g + 4 lk .. + 3 of row 32 blk + li -- exactly the
// float4 that lane (li, lk) of the MFMA reads for k-group g (consecutive lanes, consecutive 16 bytes: conflict-free ds_read_b128).
// Rows beyond M / N fetch a valid row (their products are never stored); k beyond K fetches a page of zeros.
// WI x WJ: the MFMA tiles of a wave, i.e. a block tile of 64 WI x 64 WJ.  <2, 2> is the 128 x 128 tile; <2, 3> (128 x 192, 80 KB of LDS: still two
// workgroups per CU, 24 MFMAs for five ds_read_b128) exists because a product's tile count rarely divides by the 512 workgroups the chip
// holds: zoom's first product at config 3 is 1560 tiles of 128 x 128 (3.05 rounds -- the fourth round nearly empty) and 1020 of 128 x 192
// (1.99 rounds).  launch_dma takes the shape with the fewest rounds x tile area.
// (One workgroup for the three colour channels of a tile -- three accumulator sets, whole pixels stored -- was built and measured slower,
// 396 registers and one workgroup per CU: profiles/r04_gemm.txt.  zoom's second product now has no interleaved store at all: dspfft_zoom_product.)
template <int WI, int WJ>
__global__ void __launch_bounds__(256, 2) gemm_nt_f32_mfma_dma(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                                int M, int N, int K, long long lda, long long ldb, long long ldc, int cs,
                                                                long long sa, long long sb, long long sc, float alpha, const float *zero_page, int ntn, int ntiles,
                                                                int nb1 = 0, long long sa2 = 0, long long sb2 = 0, long long sc2 = 0)
{
	constexpr int BKD = 32, G = BKD / 8, BLK = 32 * BKD;             // floats of a 32-row block of an operand tile
	constexpr int TM = 64 * WI, TN = 64 * WJ, ABLK = 2 * WI, BBLK = 2 * WJ;
	extern __shared__ __attribute__((aligned(16))) float smem[];     // [2 buffers][A blocks | B blocks]
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1;
	const int batch = (int)blockIdx.z;
	{	// blockIdx.z = b1 + nb1 * b2: two batch levels (see gemm_nt_f32_mfma), nb1 = 0: one level
		const int b2 = nb1 ? batch / nb1 : 0, b1 = batch - b2 * nb1;
		A += (long long)b1 * sa + (long long)b2 * sa2; B += (long long)b1 * sb + (long long)b2 * sb2; C += (long long)b1 * sc + (long long)b2 * sc2;
	}
	const int li = lane & 31, lk = lane >> 5;
	const int swave = __builtin_amdgcn_readfirstlane(wave);
	constexpr int NA = (ABLK + 3) / 4, NBB = (BBLK + 3) / 4;
	float *const buf0 = smem, *const buf1 = smem + (ABLK + BBLK) * BLK;
	const int nk = (K + BKD - 1) / BKD;

	// DMA: wave w stages the 32-row blocks w, w + 4, .. of A and of B: G instructions per block and K-tile.  An instruction's address is a
	// wave-uniform 64-bit base (the tile's first row, advanced by k0: scalar registers) plus a 32-bit lane offset that never changes (row and
	// k-half of the lane) plus an immediate (the k-group); its LDS destination is scalar too (the wave index is read into an SGPR) -- no vector
	// arithmetic per instruction.  (Computing full 64-bit lane pointers, with a select against the zero page, cost 55 vector instructions
	// per wave and K-tile next to its 64 MFMAs.)  Only the last K-tile of a K that is no multiple of 32 takes the selects.
	struct Aim { unsigned aoff[NA], boff[NBB]; const char *abase, *bbase; int bm, bn; };
	auto aim = [&](int tile) __attribute__((always_inline)) {
		Aim t;
		const int tm = tile / ntn, tn = tile - tm * ntn;
		t.bm = tm * TM; t.bn = tn * TN;
#pragma unroll
		for (int q = 0; q < NA; q++) { const int r = t.bm + 32 * (swave + 4 * q) + li; t.aoff[q] = (unsigned)(((long long)((r < M ? r : M - 1) - t.bm) * lda + 4 * lk) * 4); }
#pragma unroll
		for (int q = 0; q < NBB; q++) { const int r = t.bn + 32 * (swave + 4 * q) + li; t.boff[q] = (unsigned)(((long long)((r < N ? r : N - 1) - t.bn) * ldb + 4 * lk) * 4); }
		t.abase = reinterpret_cast<const char *>(A + (long long)t.bm * lda); t.bbase = reinterpret_cast<const char *>(B + (long long)t.bn * ldb);
		return t;
	};
	auto stage_any = [&](const Aim &t, float *base, int k0) __attribute__((always_inline)) {
		const char *ak = t.abase + (long long)k0 * 4, *bk = t.bbase + (long long)k0 * 4;
		if (k0 + BKD <= K) {
#pragma unroll
			for (int g = 0; g < G; g++) {
#pragma unroll
				for (int q = 0; q < NA; q++)
					if (ABLK % 4 == 0 || swave + 4 * q < ABLK)
						__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ak + t.aoff[q] + 32 * g), (__attribute__((address_space(3))) void *)(base + (swave + 4 * q) * BLK + g * 256), 16, 0, 0);
#pragma unroll
				for (int q = 0; q < NBB; q++)
					if (BBLK % 4 == 0 || swave + 4 * q < BBLK)
						__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bk + t.boff[q] + 32 * g), (__attribute__((address_space(3))) void *)(base + (ABLK + swave + 4 * q) * BLK + g * 256), 16, 0, 0);
			}
		} else {                                      // k beyond K reads the page of zeros
#pragma unroll
			for (int g = 0; g < G; g++) {
				const bool in = k0 + 8 * g + 4 * lk < K;
#pragma unroll
				for (int q = 0; q < NA; q++)
					if (ABLK % 4 == 0 || swave + 4 * q < ABLK) {
						const char *pa = in ? ak + t.aoff[q] + 32 * g : reinterpret_cast<const char *>(zero_page);
						__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)pa, (__attribute__((address_space(3))) void *)(base + (swave + 4 * q) * BLK + g * 256), 16, 0, 0);
					}
#pragma unroll
				for (int q = 0; q < NBB; q++)
					if (BBLK % 4 == 0 || swave + 4 * q < BBLK) {
						const char *pb = in ? bk + t.boff[q] + 32 * g : reinterpret_cast<const char *>(zero_page);
						__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)pb, (__attribute__((address_space(3))) void *)(base + (ABLK + swave + 4 * q) * BLK + g * 256), 16, 0, 0);
					}
			}
		}
	};

	// A workgroup takes tiles blockIdx.x, + gridDim.x, .. -- normally just the one (launch_dma).  With more, a tile's LAST K-tile sends out the
	// first operands of the workgroup's next tile, which then starts on data that is there while its predecessor's stores drain under its
	// MFMAs.  Everything a tile's K-loop uses is a constant of that loop iteration (`cur`); the next tile's addresses are worked out inside
	// the one K-tile that needs them (a first version kept them in variables the K-loop also read: 44 more registers and -5 %).
	int tile = blockIdx.x;
	{
		const Aim first = aim(tile);
		stage_any(first, buf0, 0);
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	bool flip = false;
	for (;;) {
		const Aim cur = aim(tile);
		const int next = tile + (int)gridDim.x, then = next < ntiles ? next : -1;
		f32x16 acc[WI][WJ];
		for (int i = 0; i < WI; i++) for (int j = 0; j < WJ; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
		// one K-tile: the DMA of the next one into `nxt`, the MFMAs of this one out of `now`.  The two are `__restrict__` parameters of an inlined
		// function so that the compiler knows the DMA's LDS writes and the ds_reads never meet: without that it puts s_waitcnt vmcnt(0)
		// between the DMA and the first ds_read -- the "prefetch" then lands before the tile's first MFMA instead of behind its last.
		auto ktile = [&](const float *__restrict__ now, float *__restrict__ nxt, int k0, bool more) __attribute__((always_inline)) {
			if (more) stage_any(cur, nxt, k0 + BKD);
			else if (then >= 0) { const Aim t = aim(then); stage_any(t, nxt, 0); }
			float4 a[2][WI], b[2][WJ];
			auto frags = [&](int g, int set) __attribute__((always_inline)) {
#pragma unroll
				for (int i = 0; i < WI; i++) a[set][i] = *reinterpret_cast<const float4 *>(now + (WI * wm + i) * BLK + g * 256 + lane * 4);
#pragma unroll
				for (int j = 0; j < WJ; j++) b[set][j] = *reinterpret_cast<const float4 *>(now + (ABLK + WJ * wn + j) * BLK + g * 256 + lane * 4);
			};
			// the fragments of k-group g + 1 are read while the MFMAs of group g run (two register sets).  (s_setprio around the MFMAs: -10 %.)
			frags(0, 0);
#pragma unroll
			for (int g = 0; g < G; g++) {
				const int set = g & 1;
				if (g + 1 < G) frags(g + 1, set ^ 1);
#pragma unroll
				for (int s = 0; s < 4; s++)
#pragma unroll
					for (int i = 0; i < WI; i++)
#pragma unroll
						for (int j = 0; j < WJ; j++) {
							const float av = s == 0 ? a[set][i].x : s == 1 ? a[set][i].y : s == 2 ? a[set][i].z : a[set][i].w;
							const float bv = s == 0 ? b[set][j].x : s == 1 ? b[set][j].y : s == 2 ? b[set][j].z : b[set][j].w;
							acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
						}
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next K-tile has landed (issued a K-tile of MFMAs ago)
			__syncthreads();
		};
		auto whole_tile = [&](float *__restrict__ b0, float *__restrict__ b1) __attribute__((always_inline)) {
			for (int kt = 0; kt < nk; kt += 2) {
				ktile(b0, b1, kt * BKD, kt + 1 < nk);
				if (kt + 1 < nk) ktile(b1, b0, (kt + 1) * BKD, kt + 2 < nk);
			}
		};
		if (flip) whole_tile(buf1, buf0); else whole_tile(buf0, buf1);
		// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
		for (int i = 0; i < WI; i++)
			for (int j = 0; j < WJ; j++) {
				const int n = cur.bn + wn * (32 * WJ) + j * 32 + li;
				if (n >= N) continue;
				for (int r = 0; r < 16; r++) {
					const int m = cur.bm + wm * (32 * WI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
					if (m < M) C[(long long)m * ldc + (long long)n * cs] = alpha * acc[i][j][r];
				}
			}
		if (then < 0) break;
		tile = next;
		if (nk & 1) flip = !flip;         // an odd number of K-tiles leaves the next tile's first one in the other buffer
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
// koff shifts the basis index (forward: bi = k, applybasis.c:416-418); noff the sample index (--inverse: bi = n, so the function is
// evaluated at (n + off) P + s while the pixel read stays at n P + s)
__global__ void ab_basis_kernel(float *re, float *im, int func, int ortho, long long terms, long long koff, long long noff, unsigned long long N)
{
	const size_t total = (size_t)terms * N;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const long long k = (long long)(i / N), n = (long long)(i - (size_t)k * N);
		double r, m; ab_basis(func, k + koff, n + noff, N, ortho, r, m);
		re[i] = (float)r; if (im) im[i] = (float)m;
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

static int device_cu_count(int dev)
{
	static thread_local int cus[32] = {};
	if (dev < 0 || dev >= 32) dev = 0;
	if (!cus[dev]) {
		int n = 0;
		cus[dev] = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : 256;
	}
	return cus[dev];
}

// fewer 128 x 128 tiles than the chip has CUs (DSPFFT_GEMM_TILE=128 / 64 forces either)
static bool small_product(const dim3 &grid128)
{
	static const int force = getenv("DSPFFT_GEMM_TILE") ? atoi(getenv("DSPFFT_GEMM_TILE")) : 0;
	if (force == 128) return false;
	if (force == 64) return true;
	return (long long)grid128.x * grid128.y * grid128.z < 256;
}

// operands by LDS-DMA in K-tiles of 32 (round 4) where every 16-byte piece is aligned; DSPFFT_GEMM_DMA=0: the register-staged kernel (A/B runs).
// Returns 1 when the product does not qualify (the caller launches the register-staged kernel), else 0 / an error.
static int launch_dma(const float *A, const float *B, float *C, int M, int N, int K, long long lda, long long ldb, long long ldc, int cs,
                      int batch, long long sa, long long sb, long long sc, float alpha, int nb1, long long sa2, long long sb2, long long sc2, const dim3 &grid, void *stream, bool small = false)
{
	static const int dma = getenv("DSPFFT_GEMM_DMA") ? atoi(getenv("DSPFFT_GEMM_DMA")) : 1;
	// (the K tail is masked per 16-byte piece, so K itself must be a multiple of 4: a K = 130 view of 132-wide rows would sum columns 130, 131 too)
	if (!(dma && K >= 32 && (K & 3) == 0 && ((lda | ldb | sa | sb | sa2 | sb2) & 3) == 0 && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0 && lda > 0 && ldb > 0 && lda < (1 << 22) && ldb < (1 << 22))) return 1;      // (32-bit lane offsets within a tile)
	static thread_local float *zero_page[32] = {};
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
	if (!zero_page[dev]) {
		if (hipMalloc((void **)&zero_page[dev], 256) != hipSuccess || hipMemset(zero_page[dev], 0, 256) != hipSuccess) { snprintf(g_zerr, sizeof g_zerr, "no memory for the zero page"); return -3; }
	}
	// tile shape: fewest rounds of 2 workgroups x CUs, weighted by the tile's area (DSPFFT_GEMM_SHAPE=0 / 1 forces 128 x 128 / 128 x 192)
	static const int force = getenv("DSPFFT_GEMM_SHAPE") ? atoi(getenv("DSPFFT_GEMM_SHAPE")) : -1;
	const long long slots = 2ll * device_cu_count(dev);
	auto rounds_area = [&](int tm, int tn) {
		const long long tiles = (long long)((M + tm - 1) / tm) * ((N + tn - 1) / tn) * batch;
		return ((tiles + slots - 1) / slots) * tm * tn;
	};
	// shape 0 / 1 / 2: 128 x 128, 128 x 192, 64 x 64 (products that cannot give every CU a 128 x 128 tile: `small`)
	const int shape = small ? 2 : (force >= 0 ? force == 1 : rounds_area(128, 192) < rounds_area(128, 128)) ? 1 : 0;
	const int TM = shape == 2 ? 64 : 128, TN = shape == 2 ? 64 : shape == 1 ? 192 : 128;
	const int ntn = (N + TN - 1) / TN, ntiles = ntn * ((M + TM - 1) / TM);
	const size_t lds = 2 * (size_t)(TM + TN) * 32 * sizeof(float);
	typedef void (*kern_t)(const float *, const float *, float *, int, int, int, long long, long long, long long, int, long long, long long, long long, float, const float *, int, int, int, long long, long long, long long);
	const kern_t kern = shape == 2 ? static_cast<kern_t>(gemm_nt_f32_mfma_dma<1, 1>) : shape == 1 ? static_cast<kern_t>(gemm_nt_f32_mfma_dma<2, 3>) : static_cast<kern_t>(gemm_nt_f32_mfma_dma<2, 2>);
	static thread_local bool attr[32][3] = {};            // the attribute is per device (see DevOnce, spec_kernels.h)
	if (!attr[dev][shape]) {
		if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { snprintf(g_zerr, sizeof g_zerr, "cannot raise the LDS limit"); return -4; }
		attr[dev][shape] = true;
	}
	// One workgroup per tile.  DSPFFT_GEMM_WALK=1: one product of more tiles than the chip holds workgroups gets that many workgroups, each walking its
	// tiles with the next tile's first operands in flight -- measured a little SLOWER than letting the dispatcher hand out tiles (8192 x 8192 x 4096:
	// 131.7 against 134.0 TF, config 3's second product 1.771 against 1.765 ms, same box: profiles/r04_gemm.txt), so it is off.
	static const int walk = getenv("DSPFFT_GEMM_WALK") ? atoi(getenv("DSPFFT_GEMM_WALK")) : 0;
	const long long wslots = shape == 2 ? 4 * slots / 2 : slots;
	const int gx = walk && batch == 1 && (long long)ntiles > wslots ? (int)wslots : ntiles;
	hipLaunchKernelGGL(kern, dim3(gx, 1, batch), dim3(256), lds, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, cs, sa, sb, sc, alpha, zero_page[dev], ntn, ntiles, nb1, sa2, sb2, sc2);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int dspfft_gemm_nt_f32(const float *A, const float *B, float *C, int M, int N, int K,
                                  long long lda, long long ldb, long long ldc, int cs,
                                  int batch, long long sa, long long sb, long long sc, float alpha, void *stream)
{
	if (!A || !B || !C || M < 1 || N < 1 || K < 1 || batch < 1 || cs < 1) { snprintf(g_zerr, sizeof g_zerr, "bad arguments"); return -1; }
	dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
	static const int small_dma = getenv("DSPFFT_GEMM_SMALL_DMA") ? atoi(getenv("DSPFFT_GEMM_SMALL_DMA")) : 1;
	if (small_product(grid)) {
		if (small_dma) if (int rc = launch_dma(A, B, C, M, N, K, lda, ldb, ldc, cs, batch, sa, sb, sc, alpha, 0, 0, 0, 0, grid, stream, true); rc != 1) return rc;
		dim3 g64((N + 63) / 64, (M + 63) / 64, batch);
		if (K >= 16) hipLaunchKernelGGL((gemm_nt_f32_mfma<16, 1>), g64, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, cs, sa, sb, sc, alpha, 0, 0ll, 0ll, 0ll);
		else hipLaunchKernelGGL((gemm_nt_f32_mfma<8, 1>), g64, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, cs, sa, sb, sc, alpha, 0, 0ll, 0ll, 0ll);
		return hipGetLastError() == hipSuccess ? 0 : -4;
	}
	if (int rc = launch_dma(A, B, C, M, N, K, lda, ldb, ldc, cs, batch, sa, sb, sc, alpha, 0, 0, 0, 0, grid, stream); rc != 1) return rc;
	static const int bk = getenv("DSPFFT_GEMM_BK") ? atoi(getenv("DSPFFT_GEMM_BK")) : 8;
	if (K >= 32 && bk == 32) hipLaunchKernelGGL(gemm_nt_f32_mfma<32>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, cs, sa, sb, sc, alpha);
	else if (K >= 16 && bk >= 16) hipLaunchKernelGGL(gemm_nt_f32_mfma<16>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, cs, sa, sb, sc, alpha);
	else hipLaunchKernelGGL(gemm_nt_f32_mfma<8>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, cs, sa, sb, sc, alpha);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

// the same product over TWO batch levels in one launch: batch index (b1, b2), offsets b1 * s?1 + b2 * s?2
static int gemm_nt_f32_batch2(const float *A, const float *B, float *C, int M, int N, int K, long long lda, long long ldb, long long ldc,
                              int nb1, long long sa1, long long sb1, long long sc1, int nb2, long long sa2, long long sb2, long long sc2, void *stream)
{
	dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, nb1 * nb2);
	static const int small_dma = getenv("DSPFFT_GEMM_SMALL_DMA") ? atoi(getenv("DSPFFT_GEMM_SMALL_DMA")) : 1;
	if (small_product(grid)) {
		if (small_dma) if (int rc = launch_dma(A, B, C, M, N, K, lda, ldb, ldc, 1, nb1 * nb2, sa1, sb1, sc1, 1.f, nb1, sa2, sb2, sc2, grid, stream, true); rc != 1) return rc;
		dim3 g64((N + 63) / 64, (M + 63) / 64, nb1 * nb2);
		if (K >= 16) hipLaunchKernelGGL((gemm_nt_f32_mfma<16, 1>), g64, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, 1, sa1, sb1, sc1, 1.f, nb1, sa2, sb2, sc2);
		else hipLaunchKernelGGL((gemm_nt_f32_mfma<8, 1>), g64, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, 1, sa1, sb1, sc1, 1.f, nb1, sa2, sb2, sc2);
		return hipGetLastError() == hipSuccess ? 0 : -4;
	}
	// the LDS-DMA kernel from one full round of the chip up (applybasis' 2048^2 spectrum, 768 tiles: 1.01 ms on it against 1.25 ms on this one)
	static const long long dma_min = getenv("DSPFFT_GEMM_DMA_MIN_TILES") ? atoll(getenv("DSPFFT_GEMM_DMA_MIN_TILES")) : 512;
	if ((long long)grid.x * grid.y * grid.z >= dma_min)
		if (int rc = launch_dma(A, B, C, M, N, K, lda, ldb, ldc, 1, nb1 * nb2, sa1, sb1, sc1, 1.f, nb1, sa2, sb2, sc2, grid, stream); rc != 1) return rc;
	if (K >= 16) hipLaunchKernelGGL(gemm_nt_f32_mfma<16>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, 1, sa1, sb1, sc1, 1.f, nb1, sa2, sb2, sc2);
	else hipLaunchKernelGGL(gemm_nt_f32_mfma<8>, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, 1, sa1, sb1, sc1, 1.f, nb1, sa2, sb2, sc2);
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
	// Tt (vw x 3 x ch: row 3 x + z) = XB (vw x cw) . plane_z[:ch, :cw]^T: the channels interleave by ROW, so every store is a contiguous run of ch floats
	// (all rows kept, ch == h: the three planes are one matrix of 3 h rows and the product is one launch -- 1020 tiles of 128 x 192 for config 3
	// instead of 3 x 540 of 128 x 128, two full rounds of the chip instead of three and a sixth)
	int rc = ch == (size_t)h ? dspfft_gemm_nt_f32(d_xb, planes, Tt, vw, 3 * h, (int)cw, (long long)cw, w, (long long)ch * 3, 1, 1, 0, 0, 0, 1.f, stream)
	                         : dspfft_gemm_nt_f32(d_xb, planes, Tt, vw, (int)ch, (int)cw, (long long)cw, w, (long long)ch * 3, 1, 3, 0, (long long)npix, (long long)ch, 1.f, stream);
	if (rc) return rc;
	// out (vh x 3 vw: the interleaved frame as it is) = YB (vh x ch) . Tt^T / (w h): ONE product for the three channels, plain contiguous stores
	return dspfft_gemm_nt_f32(d_yb, Tt, d_out, vh, vw * 3, (int)ch, (long long)ch, (long long)ch, (long long)vw * 3, 1, 1, 0, 0, 0, 1.f / ((float)w * (float)h), stream);
}

// ---- applybasis' partial sums (applybasis/applybasis.c:410-431) as TWO batched NT GEMM launches ----
// out[kh][kw][nh][nw][j] = sum_{sh, sw} f_h(kh + offh, nh Ph + sh) f_w(kw + offw, nw Pw + sw) pix_j[nh Ph + sh][nw Pw + sw]
// (--inverse: the offset belongs to the block index, f(k, (n + off) P + s), the pixel read does not move: applybasis.c:372-378,416-420)
// Forward (K = terms, N = image / partsum) and --inverse (K = image size, N = terms / partsum, applybasis.c:378-389) are the
// same sums with different (K, N); pixels may be complex (a .coeff input, applybasis.c:319-338).
//   step 1  Tt[j][(part, kw)][nw][y]            = sum_sw Fw[(part, kw)][nw Pw + sw] pix_j[y][nw Pw + sw]        batch (nw, j)
//   step 2  P[j][(hp, kh)][nh][(tp, kw, nw)]    = sum_sh Fh[(hp, kh)][nh Ph + sh] Tt[j][(tp, kw)][nw][nh Ph + sh]  batch (nh, j)
// with the real and imaginary basis parts stacked as extra rows, so each step is ONE launch whatever the function
// (round 1: 6 to 24 launches from a host loop, MfmaUtil 0.02 on a 512 x 512 image).
__global__ void ab_combine2_kernel(float *out, const float *P, int cplx, int Kh, int Kw, int Nh, int Nw, int accumulate_as_imag)
{
	const int C = cplx ? 2 : 1;
	const size_t row = (size_t)C * Kw * Nw;                     // floats per (hp, kh, nh)
	const size_t perj = (size_t)C * Kh * Nh * row;
	const size_t total = (size_t)Kh * Kw * Nh * Nw * 3;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const size_t j = i % 3, t = i / 3;
		const size_t nw = t % Nw, nh = (t / Nw) % Nh, kw = (t / ((size_t)Nw * Nh)) % Kw, kh = t / ((size_t)Nw * Nh * Kw);
		auto at = [&](int hp, int tp) { return P[j * perj + (((size_t)hp * Kh + kh) * Nh + nh) * row + ((size_t)tp * Kw + kw) * Nw + nw]; };
		float re, im;
		if (cplx) { re = at(0, 0) - at(1, 1); im = at(0, 1) + at(1, 0); } else { re = at(0, 0); im = 0.f; }
		float *o = out + i * 2;
		if (accumulate_as_imag) { o[0] -= im; o[1] += re; }       // + i * (re + i im): the imaginary part of complex pixels
		else { o[0] = re; o[1] = im; }
	}
}

// ---- small partial-sum blocks (-P 8x8, 16x16 ...: applybasis/applybasis.c:410-431) in ONE launch (round 5) ----
// With blocks of P x P pixels the two products above have an inner dimension of P: the generic GEMM runs them on its register-staged K < 32
// path, writes the intermediate [j][(hp, kh)][nh][(tp, kw, nw)], and a third kernel re-reads it to interleave the result -- the result's bytes
// cross HBM three times (0.41 ms for a 403 MB result: 0.98 TB/s, profiles/r05_applybasis_blocks.json).  The arithmetic is nothing (0.9 GFLOP);
// what matters is writing the result once, in runs.  One workgroup owns block row nh and a 16 x 16 tile of basis functions (kh, kw) and walks the
// row in tiles of 16 blocks:
//   stage   the tile's pixels, P rows x 16 P pixels, into LDS (whole rows of the image: coalesced)
//   A       T[kw][sh][c] = sum_sw Fw[kw][nw P + sw] pix[nh P + sh][nw P + sw][j],  c = 3 nw + j: per block a (16 kw x P) . (P x 3 P) product on the matrix cores, into LDS
//   B       out[kh][kw][nh][c] = sum_sh Fh[kh][nh P + sh] T[kw][sh][c] on the matrix cores: v_mfma_f32_16x16x4_f32 tiles of 16 kh x 16 c, K = P in steps
//           of 4 (A operand: the lane's basis value, kept in a register for the whole row; B operand: one LDS word), four kw per wave;
//           the accumulator's four registers are four kh rows x 16 consecutive c = 128 contiguous bytes per row of the complex result -- stored as
//           (re, im) pairs straight from the accumulator, no intermediate, no combine pass.
// Complex bases (dft / idft): T and the accumulators in two parts; the imaginary pass of complex pixels (a .coeff input) adds i x the result in place.
typedef float ab_f4 __attribute__((ext_vector_type(4)));
// khspan: the workgroup walks `khspan` tiles of kh for its (nh, kw tile), so that T is computed once for all of them (16 x 16 blocks: phase A is 768
// multiply-adds and as many LDS reads per thread, as long as the 98 KB one kh tile stores)
template <int P, bool CPLX>
__global__ void __launch_bounds__(256) ab_blocks_kernel(float *out, const float *pix, const float *Fw, const float *Fh, int w, int h, int Kw, int Kh, int Nw, int Nh, int as_imag, int khspan)
{
	constexpr int NB = 16, CT = NB * 3, XW = NB * P * 3, NC = CPLX ? 2 : 1, KS = P / 4;
	extern __shared__ __attribute__((aligned(16))) float ab_lds[];
	float *pixs = ab_lds;                               // [P][XW + 4]
	float *Ts = ab_lds + P * (XW + 4);                  // [NC][16][P][CT]
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int nh = blockIdx.x, kwb = blockIdx.y * 16;
	// phase B's A operands: A[i = lane & 15][k = lane >> 4] of K-step s = Fh[khb + i][nh P + 4 s + k]
	float ah[NC][KS];
	auto load_ah = [&](int khb) {
		const int kh = khb + (lane & 15);
		for (int s_ = 0; s_ < KS; s_++)
			for (int part = 0; part < NC; part++)
				ah[part][s_] = kh < Kh ? Fh[((size_t)part * Kh + kh) * h + (size_t)nh * P + 4 * s_ + (lane >> 4)] : 0.f;
	};
	const int kh_first = (int)blockIdx.z * khspan * 16, kh_end = kh_first + khspan * 16 < Kh ? kh_first + khspan * 16 : Kh;
	if (khspan == 1) load_ah(kh_first);
	// phase A on the matrix cores as well: per block, T[kw][(sh, j)] = Fw_blk (16 kw x P) . Pix_blk (P x 3 P): 16 x 16 tiles over the 3 P columns (sh, j)
	// (24 columns of an 8 x 8 block fill one tile and a half), K = P in steps of 4; a wave takes four of the tile's sixteen blocks.  The lane's
	// Fw values (A operands) are fetched before the pixels are waited for.
	constexpr int NTA = (3 * P + 15) / 16, XWP = XW + 4;      // pixel rows padded by four floats: the 16 columns of a tile then fall on 16 banks
	for (int nwb = 0; nwb < Nw; nwb += NB) {
		float af[NC][4][KS];
		for (int b = 0; b < 4; b++) {
			const int nw = nwb + wave * 4 + b, kw = kwb + (lane & 15);
			for (int s_ = 0; s_ < KS; s_++)
				for (int part = 0; part < NC; part++)
					af[part][b][s_] = (kw < Kw && nw < Nw) ? Fw[((size_t)part * Kw + kw) * w + (size_t)nw * P + 4 * s_ + (lane >> 4)] : 0.f;
		}
		const int nx = (Nw - nwb < NB ? Nw - nwb : NB) * P * 3;          // floats of this tile's pixel rows that exist
		for (int e = tid; e < P * XW; e += 256) {
			const int sh = e / XW, xo = e - sh * XW;
			pixs[sh * XWP + xo] = xo < nx ? pix[((size_t)(nh * P + sh) * w + (size_t)nwb * P) * 3 + xo] : 0.f;
		}
		__syncthreads();
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const int nw_l = wave * 4 + b;
#pragma unroll
			for (int nt = 0; nt < NTA; nt++) {
				const int q = 16 * nt + (lane & 15), sh = q / 3, jj = q - 3 * sh;      // this lane's column (sh, j) of the block's 3 P
				const bool col = q < 3 * P;
				ab_f4 tr = {0.f, 0.f, 0.f, 0.f}, ti = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int s_ = 0; s_ < KS; s_++) {
					const float bv = col ? pixs[sh * XWP + (nw_l * P + 4 * s_ + (lane >> 4)) * 3 + jj] : 0.f;
					tr = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][b][s_], bv, tr, 0, 0, 0);
					if constexpr (CPLX) ti = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][b][s_], bv, ti, 0, 0, 0);
				}
				if (col) {
#pragma unroll
					for (int r = 0; r < 4; r++) {
						const int kw_l = 4 * (lane >> 4) + r;
						Ts[((size_t)kw_l * P + sh) * CT + nw_l * 3 + jj] = tr[r];
						if constexpr (CPLX) Ts[(((size_t)16 + kw_l) * P + sh) * CT + nw_l * 3 + jj] = ti[r];
					}
				}
			}
		}
		__syncthreads();
		for (int khb = kh_first; khb < kh_end; khb += 16) {
		if (khspan > 1) load_ah(khb);
		for (int q = 0; q < 4; q++) {
			const int kw_l = wave * 4 + q, kw = kwb + kw_l;
			if (kw >= Kw) break;                                 // (uniform per wave)
			for (int nt = 0; nt < CT / 16; nt++) {
				ab_f4 re = {0.f, 0.f, 0.f, 0.f}, im = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int s_ = 0; s_ < KS; s_++) {
					const float br = Ts[((size_t)kw_l * P + 4 * s_ + (lane >> 4)) * CT + 16 * nt + (lane & 15)];
					re = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[0][s_], br, re, 0, 0, 0);
					if constexpr (CPLX) {
						const float bi = Ts[(((size_t)16 + kw_l) * P + 4 * s_ + (lane >> 4)) * CT + 16 * nt + (lane & 15)];
						re = __builtin_amdgcn_mfma_f32_16x16x4f32(-ah[1][s_], bi, re, 0, 0, 0);
						im = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[0][s_], bi, im, 0, 0, 0);
						im = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[1][s_], br, im, 0, 0, 0);
					}
				}
				const int c = nwb * 3 + 16 * nt + (lane & 15);
				if (c < Nw * 3) {
#pragma unroll
					for (int r = 0; r < 4; r++) {
						const int kh = khb + 4 * (lane >> 4) + r;
						if (kh < Kh) {
							float2 *o = reinterpret_cast<float2 *>(out) + (((size_t)kh * Kw + kw) * Nh + nh) * (size_t)Nw * 3 + c;
							float2 v;
							if (as_imag) { v = *o; v.x -= im[r]; v.y += re[r]; }      // + i (re + i im): the imaginary part of complex pixels
							else { v.x = re[r]; v.y = im[r]; }
							*o = v;
						}
					}
				}
			}
		}
		}
		__syncthreads();
	}
}
template <int P, bool CPLX>
static int launch_ab_blocks(float *out, const float *pix, const float *Fw, const float *Fh, int w, int h, int Kw, int Kh, int Nw, int Nh, int as_imag, hipStream_t s)
{
	const size_t lds = ((size_t)P * (16 * P * 3 + 4) + (size_t)(CPLX ? 2 : 1) * 16 * P * 48) * sizeof(float);
	static thread_local bool attr[32] = {};
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
	if (lds > 48 * 1024 && !attr[dev]) {
		if (hipFuncSetAttribute(reinterpret_cast<const void *>(ab_blocks_kernel<P, CPLX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -4;
		attr[dev] = true;
	}
	// kh tiles per workgroup: all of them for 16 x 16 blocks (phase A weighs as much as a kh tile's stores: 1024^2 image, 64 x 64 terms 0.62 ms with one tile
	// per workgroup, 0.36 with two, 0.255 with all four -- also where that leaves fewer workgroups than CUs: 512^2 0.168 / 0.115 ms), one for smaller blocks
	const int ktiles = (Kh + 15) / 16;
	static const int span_env = getenv("DSPFFT_AB_KHSPAN") ? atoi(getenv("DSPFFT_AB_KHSPAN")) : 0;      // experiments
	const int khspan = span_env > 0 ? (span_env < ktiles ? span_env : ktiles) : (P >= 16 ? ktiles : 1);
	hipLaunchKernelGGL((ab_blocks_kernel<P, CPLX>), dim3(Nh, (Kw + 15) / 16, (ktiles + khspan - 1) / khspan), dim3(256), lds, s, out, pix, Fw, Fh, w, h, Kw, Kh, Nw, Nh, as_imag, khspan);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}
// which block sizes the one-launch kernel takes (square, 4 / 8 / 16 pixels); DSPFFT_AB_BLOCKS=0: never (A/B runs)
static bool ab_blocks_applies(int Pw, int Ph, int Kw, int Kh)
{
	static const int on = getenv("DSPFFT_AB_BLOCKS") ? atoi(getenv("DSPFFT_AB_BLOCKS")) : 1;
	return on && Pw == Ph && (Pw == 4 || Pw == 8 || Pw == 16) && (Kw + 15) / 16 <= 65535 && (Kh + 15) / 16 <= 65535;
}

extern "C" size_t dspfft_applybasis_work_floats_ex(int w, int h, int Kw, int Kh, int Nw, int Nh, int func)
{
	const size_t C = func <= 1 ? 2 : 1;
	return (size_t)3 * w * h + C * ((size_t)Kw * w + (size_t)Kh * h) + C * 3 * (size_t)Kw * Nw * h + C * C * 3 * (size_t)Kh * Nh * Kw * Nw;
}
extern "C" size_t dspfft_applybasis_work_floats(int w, int h, int Kw, int Kh, int Pw, int Ph, int func)
{
	return dspfft_applybasis_work_floats_ex(w, h, Kw, Kh, w / Pw, h / Ph, func);
}

static int applybasis_core(float *d_out, const float *d_pixels, int w, int h, int func, int ortho, int Kw, int Kh, int Nw, int Nh, int Pw, int Ph,
                           long long offw, long long offh, int inverse, float *d_work, int as_imag, hipStream_t s)
{
	const long long kow = inverse ? 0 : offw, koh = inverse ? 0 : offh, now = inverse ? offw * Pw : 0, noh = inverse ? offh * Ph : 0;
	const int cplx = func <= 1, C = cplx ? 2 : 1;
	const size_t npix = (size_t)w * h;
	float *planes = d_work;
	float *Fw = planes + 3 * npix, *Fh = Fw + (size_t)C * Kw * w;               // [(part, k)][n]: imaginary rows stacked under the real ones
	float *Tt = Fh + (size_t)C * Kh * h;                                         // [j][(part, kw)][nw][y]
	float *P = Tt + (size_t)3 * C * Kw * Nw * h;                                 // [j][(hp, kh)][nh][(tp, kw, nw)]
	hipLaunchKernelGGL(ab_basis_kernel, dim3(512), dim3(256), 0, s, Fw, cplx ? Fw + (size_t)Kw * w : nullptr, func, ortho, (long long)Kw, kow, now, (unsigned long long)w);
	hipLaunchKernelGGL(ab_basis_kernel, dim3(512), dim3(256), 0, s, Fh, cplx ? Fh + (size_t)Kh * h : nullptr, func, ortho, (long long)Kh, koh, noh, (unsigned long long)h);
	if (ab_blocks_applies(Pw, Ph, Kw, Kh)) {             // small square blocks: the two products and the interleave in one launch, the result written once
		switch (Pw * 2 + cplx) {
		case 8: return launch_ab_blocks<4, false>(d_out, d_pixels, Fw, Fh, w, h, Kw, Kh, Nw, Nh, as_imag, s);
		case 9: return launch_ab_blocks<4, true>(d_out, d_pixels, Fw, Fh, w, h, Kw, Kh, Nw, Nh, as_imag, s);
		case 16: return launch_ab_blocks<8, false>(d_out, d_pixels, Fw, Fh, w, h, Kw, Kh, Nw, Nh, as_imag, s);
		case 17: return launch_ab_blocks<8, true>(d_out, d_pixels, Fw, Fh, w, h, Kw, Kh, Nw, Nh, as_imag, s);
		case 32: return launch_ab_blocks<16, false>(d_out, d_pixels, Fw, Fh, w, h, Kw, Kh, Nw, Nh, as_imag, s);
		default: return launch_ab_blocks<16, true>(d_out, d_pixels, Fw, Fh, w, h, Kw, Kh, Nw, Nh, as_imag, s);
		}
	}
	hipLaunchKernelGGL(deinterleave3_kernel, dim3(1024), dim3(256), 0, s, planes, d_pixels, npix);
	int rc = gemm_nt_f32_batch2(Fw, planes, Tt, C * Kw, h, Pw, w, w, (long long)Nw * h,
	                            Nw, Pw, Pw, h, 3, 0, (long long)npix, (long long)C * Kw * Nw * h, s);
	if (rc) return rc;
	const long long row = (long long)C * Kw * Nw;
	rc = gemm_nt_f32_batch2(Fh, Tt, P, C * Kh, (int)row, Ph, h, h, (long long)Nh * row,
	                        Nh, Ph, Ph, row, 3, 0, (long long)C * Kw * Nw * h, (long long)C * Kh * Nh * row, s);
	if (rc) return rc;
	hipLaunchKernelGGL(ab_combine2_kernel, dim3(1024), dim3(256), 0, s, d_out, P, cplx, Kh, Kw, Nh, Nw, as_imag);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int dspfft_applybasis_partsums_ex(float *d_out, const float *d_pix_re, const float *d_pix_im, int w, int h, int func, int ortho,
                                             int Kw, int Kh, int Nw, int Nh, int Pw, int Ph, long long offw, long long offh, int inverse, float *d_work, void *stream)
{
	if (!d_out || !d_pix_re || !d_work || func < 0 || func > 11 || Kw < 1 || Kh < 1 || Nw < 1 || Nh < 1 || Pw < 1 || Ph < 1 ||
	    (long long)Nw * Pw > w || (long long)Nh * Ph > h) {
		snprintf(g_zerr, sizeof g_zerr, "bad arguments (the partial-sum blocks N x P must fit in the image)"); return -1;
	}
	hipStream_t s = (hipStream_t)stream;
	int rc = applybasis_core(d_out, d_pix_re, w, h, func, ortho, Kw, Kh, Nw, Nh, Pw, Ph, offw, offh, inverse, d_work, 0, s);
	if (!rc && d_pix_im) rc = applybasis_core(d_out, d_pix_im, w, h, func, ortho, Kw, Kh, Nw, Nh, Pw, Ph, offw, offh, inverse, d_work, 1, s);
	return rc;
}

extern "C" int dspfft_applybasis_partsums(float *d_out, const float *d_pixels, int w, int h, int func, int ortho,
                                          int Kw, int Kh, int Pw, int Ph, long long offw, long long offh,
                                          float *d_work, void *stream)
{
	if (Pw < 1 || Ph < 1 || w % Pw || h % Ph) { snprintf(g_zerr, sizeof g_zerr, "bad arguments (the partial-sum block must divide the image)"); return -1; }
	return dspfft_applybasis_partsums_ex(d_out, d_pixels, nullptr, w, h, func, ortho, Kw, Kh, w / Pw, h / Ph, Pw, Ph, offw, offh, 0, d_work, stream);
}

// ---- the rendered frame (applybasis.c:392-442): realize -> rescale (one type, or two interpolated) -> range -> cells of
// scale x scale pixels at INDEX(d) = ((size.d * bi.d + i.d) * scale + padding * bi.d + padding), alpha = 1 ----
__device__ inline double ab_rescale(int type, double c, double scale)
{
	switch (type) {
	case 1: return copysign(log1p(fabs(c)) / log1p(scale), c);                                         // logscale
	case 2: { const double r = sqrt(scale); c /= r; return copysign(log1p(fabs(c)) / log1p(r), c); }  // gain
	case 3: c /= scale; return copysign(log1p(fabs(c)) / log1p(1.0), c);                               // loglevel
	default: return c / scale;                                                                         // linear
	}
}
__global__ void ab_render_kernel(float *frame, const float *part, int Kw, int Kh, int Nw, int Nh, int inverse, int scale, int padding,
                                 int plane, int rescale0, int rescale1, int range, double coeff_scale, double insize_wh, long long fw)
{
	const size_t total = (size_t)Kh * Kw * Nh * Nw;
	for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
		const size_t nw = t % Nw, nh = (t / Nw) % Nh, kw = (t / ((size_t)Nw * Nh)) % Kw, kh = t / ((size_t)Nw * Nh * Kw);
		double v[3];
		for (int j = 0; j < 3; j++) {
			const double re = part[(t * 3 + j) * 2], im = part[(t * 3 + j) * 2 + 1];
			double r = plane == 1 ? im : plane == 2 ? hypot(re, im) : plane == 3 ? atan2(im + 2.220446049250313e-16, re) / 3.14159265358979323846 : re;
			double c0 = ab_rescale(rescale0, r, coeff_scale);
			if (rescale1 >= 0) {
				const double c1 = ab_rescale(rescale1, r, coeff_scale), NN = sqrt(insize_wh) - 1, nn = sqrt(coeff_scale) - 1;
				c0 = ((NN - nn) * c0 + nn * c1) / NN;
			}
			v[j] = c0;
		}
		if (range == 0) for (int j = 0; j < 3; j++) v[j] = (v[j] + 1) / 2;                 // shift and shift2 (the latter's pre-mapping is the caller's)
		else if (range == 1) for (int j = 0; j < 3; j++) v[j] = fabs(v[j]);              // abs
		else if (range == 2) for (int j = 0; j < 3; j++) v[j] += v[j] < 0;               // invert
		else if (!(v[0] >= 0 && v[1] >= 0 && v[2] >= 0)) {                               // hue
			const double a = fabs(v[0]), b = fabs(v[1]), c = fabs(v[2]);
			v[0] = (-a + 2 * b + 2 * c) / 3; v[1] = (2 * a - b + 2 * c) / 3; v[2] = (2 * a + 2 * b - c) / 3;
		}
		// forward: bi = k (terms), i = n, size = N;  --inverse: bi = n, i = k, size = K (applybasis.c:378-389)
		const size_t bw = inverse ? nw : kw, bh = inverse ? nh : kh, iw = inverse ? kw : nw, ih = inverse ? kh : nh;
		const size_t sw_ = inverse ? (size_t)Kw : (size_t)Nw, sh_ = inverse ? (size_t)Kh : (size_t)Nh;
		const size_t x0 = (sw_ * bw + iw) * scale + (size_t)padding * bw + padding, y0 = (sh_ * bh + ih) * scale + (size_t)padding * bh + padding;
		for (int ys = 0; ys < scale; ys++)
			for (int xs = 0; xs < scale; xs++) {
				float *o = frame + ((y0 + ys) * (size_t)fw + x0 + xs) * 4;
				o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = 1.f;
			}
	}
}
__global__ void ab_fill_kernel(float *frame, size_t npix, float r, float g, float b, float a)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
		frame[4 * i] = r; frame[4 * i + 1] = g; frame[4 * i + 2] = b; frame[4 * i + 3] = a;
	}
}

extern "C" int dspfft_applybasis_render(float *d_frame, const float *d_partsums, int Kw, int Kh, int Nw, int Nh, int inverse, int scale, int padding,
                                        int plane, int rescale0, int rescale1, int range, double coeff_scale, double insize_wh,
                                        const float padcolor[4], void *stream)
{
	if (!d_frame || !d_partsums || Kw < 1 || Kh < 1 || Nw < 1 || Nh < 1 || scale < 1 || padding < 0 || plane < 0 || plane > 3 ||
	    rescale0 < 0 || rescale0 > 3 || rescale1 > 3 || range < 0 || range > 3 || !padcolor) { snprintf(g_zerr, sizeof g_zerr, "bad arguments"); return -1; }
	// framesize = size * terms * scale + padding * terms + padding with (size, terms) = (N, K) forward, (K, N) inverse: the same product
	const long long tw = inverse ? Nw : Kw, th = inverse ? Nh : Kh;
	const long long fw = (long long)Kw * Nw * scale + (long long)padding * tw + padding, fh = (long long)Kh * Nh * scale + (long long)padding * th + padding;
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(ab_fill_kernel, dim3(1024), dim3(256), 0, s, d_frame, (size_t)fw * fh, padcolor[0], padcolor[1], padcolor[2], padcolor[3]);
	hipLaunchKernelGGL(ab_render_kernel, dim3(1024), dim3(256), 0, s, d_frame, d_partsums, Kw, Kh, Nw, Nh, inverse, scale, padding, plane, rescale0, rescale1, range,
	                   coeff_scale, insize_wh, fw);
	return hipGetLastError() == hipSuccess ? 0 : -4;
}
