// block_core.h -- small blocks transformed along ALL their axes in one pass (motion's block modes: --blocksize 8x8x8 and the
// like, motion/motion.c:535 with block dims; JPEG-like 8x8x1 blocks), and motion's whole per-block pipeline -- load, forward
// transform, coefficient filter, inverse transform, store (motion.c:617-776) -- as one pass.
//
// The per-axis TINY passes (dct_core.h) move a volume cut into small blocks through HBM once per axis: 24 B/sample for a 3-D
// forward transform, 56 B/sample and more for the filtered roundtrip.  A block of <= 16^3 samples fits LDS many times over, so
// here one workgroup takes G blocks and does everything between one load and one store: 8 B/sample, 2 B/sample on 8-bit video.
//   phase x   thread = one x line of a block: NX contiguous samples from global memory, transform in registers (tiny_dct),
//             write to the LDS tile [NZ][NY][G * NX]                              (the inverse ends with the mirror image of this)
//   phase y/z thread = one y (z) line of the tile; neighbouring threads, neighbouring columns: conflict-free; in place
//   store     rows back to global memory, float4 per lane
// Two layouts: the blocks of a [D][H][W] volume side by side along x (block-to-block stride NX: a row of the tile is G * NX
// contiguous samples) and block-major buffers like the reference's per-block arrays (rows of a block contiguous, blocks
// NX * NY * NZ apart: a block is one contiguous run).  Plain C++17 for device (hipcc) and the CPU emulation (g++).
#pragma once
#include "dct_core.h"
#include "elementwise_core.h"
#include "motion_filter.h"

namespace dspfft {

enum { BLOCK_THREADS = 256, BLOCK_MAX_DIMS = 5 };

struct BlockGeom {
	int nx, ny, nz;               // block extents (nz = 1: 2-D blocks); x is the contiguous axis
	long long sy_in, sz_in, sy_out, sz_out;   // strides between the rows of a block
	long long sxb_in, sxb_out;    // stride between the G blocks of a workgroup
	int rows_fast;                // neighbouring threads take neighbouring ROWS of one block (block-major layouts) instead of
	                              // the same row of neighbouring blocks
	int G;                        // blocks per workgroup
	int nxb;                      // blocks along the grouped dimension
	int ngroups;                  // ceil(nxb / G)
	int pitch;                    // floats between rows of the LDS tile (G * nx)
	int nd;                       // further batch dimensions (rows of blocks, planes ...)
	int bn[BLOCK_MAX_DIMS];
	long long bis[BLOCK_MAX_DIMS], bos[BLOCK_MAX_DIMS];
	FastDiv bdiv[BLOCK_MAX_DIMS];
	FastDiv gdiv;                 // divide by ngroups
};
// scales of one plan: `scale` goes with the last axis, in0 / out0 are the per-axis index-0 factors (x, y, z)
struct BlockScales { float scale, in0[3], out0[3]; };
struct BlockArgs : BlockGeom {
	int kind;                     // KIND_* (all axes alike)
	const float *in;
	float *out;
	BlockScales s;
};
// forward -> filter -> inverse; in8 / out8 replace in / out when set (8-bit samples in the same element layout)
struct BlockRtArgs : BlockGeom {
	const float *in;
	float *out;
	const uint8_t *in8;
	uint8_t *out8;
	double mul8;
	BlockScales f, i;
	MotionFilter filt;            // filt.enabled = 0: no filter; positions are the block's own (z, y, x)
	unsigned long long *coded;
};

DSP_HD TinyArgs block_axis_args(const BlockScales &s, int axis, bool last)
{
	TinyArgs t;
	t.scale = last ? s.scale : 1.f; t.in_scale0 = s.in0[axis]; t.out_scale0 = s.out0[axis];
	return t;
}

DSP_HD void block_base(const BlockGeom &a, uint32_t wg, long long &bin, long long &bout, int &cnt)
{
	uint32_t r = a.gdiv.div_exact(wg);
	const uint32_t grp = wg - r * (uint32_t)a.ngroups;
	bin = (long long)grp * a.G * a.sxb_in; bout = (long long)grp * a.G * a.sxb_out;
	cnt = a.nxb - (int)grp * a.G;
	if (cnt > a.G) cnt = a.G;
	for (int d = 0; d < a.nd; d++) {
		const uint32_t q = a.bdiv[d].div_exact(r), i = r - q * (uint32_t)a.bn[d];
		bin += (long long)i * a.bis[d]; bout += (long long)i * a.bos[d];
		r = q;
	}
}
// x line l of the workgroup -> (tile row, block)
DSP_HD void block_line_of(const BlockGeom &a, int l, int rows, int cnt, int &row, int &g)
{
	if (a.rows_fast) { g = l / rows; row = l - g * rows; } else { row = l / cnt; g = l - row * cnt; }
}

// phase x, forward side: load (float or 8-bit) + x transform into the tile
template <int NX, int NY, int NZ, int KIND, bool U8>
DSP_HD void block_load_x(const BlockGeom &a, const TinyArgs &tx, const float *in, const uint8_t *in8, float *lds, long long bin, int cnt, int tid)
{
	const int lines = NZ * NY * cnt;
	for (int l = tid; l < lines; l += BLOCK_THREADS) {
		int row, g;
		block_line_of(a, l, NZ * NY, cnt, row, g);
		const int z = row / NY, y = row - z * NY;
		const long long off = bin + (long long)g * a.sxb_in + (long long)z * a.sz_in + (long long)y * a.sy_in;
		float x[NX], o[NX];
		if constexpr (U8) {
#pragma unroll
			for (int j = 0; j < NX / 4; j++) {
				uint32_t w4;
				__builtin_memcpy(&w4, in8 + off + 4 * j, 4);
#pragma unroll
				for (int q = 0; q < 4; q++) x[4 * j + q] = (float)((w4 >> (8 * q)) & 0xffu);
			}
		} else {
#pragma unroll
			for (int j = 0; j < NX / 4; j++) {      // 16-byte aligned: the planner only fuses such layouts
				const float4 v = reinterpret_cast<const float4 *>(in + off)[j];
				x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
			}
		}
		tiny_dct<NX, KIND>(tx, x, o);
		float4 *q = reinterpret_cast<float4 *>(lds + row * a.pitch + g * NX);
#pragma unroll
		for (int j = 0; j < NX / 4; j++) { float4 v; v.x = o[4 * j]; v.y = o[4 * j + 1]; v.z = o[4 * j + 2]; v.w = o[4 * j + 3]; q[j] = v; }
	}
}
// phase x, inverse side: x transform out of the tile + store (float or quantised 8-bit, motion.c:760-776)
template <int NX, int NY, int NZ, int KIND, bool U8>
DSP_HD void block_store_x(const BlockGeom &a, const TinyArgs &tx, float *out, uint8_t *out8, double mul8, const float *lds, long long bout, int cnt, int tid)
{
	const int lines = NZ * NY * cnt;
	for (int l = tid; l < lines; l += BLOCK_THREADS) {
		int row, g;
		block_line_of(a, l, NZ * NY, cnt, row, g);
		const int z = row / NY, y = row - z * NY;
		const long long off = bout + (long long)g * a.sxb_out + (long long)z * a.sz_out + (long long)y * a.sy_out;
		float x[NX], o[NX];
		const float4 *q = reinterpret_cast<const float4 *>(lds + row * a.pitch + g * NX);
#pragma unroll
		for (int j = 0; j < NX / 4; j++) { const float4 v = q[j]; x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w; }
		tiny_dct<NX, KIND>(tx, x, o);
		if constexpr (U8) {
#pragma unroll
			for (int j = 0; j < NX / 4; j++) {
				uint32_t w4 = 0;
#pragma unroll
				for (int k = 0; k < 4; k++) w4 |= quantise_u8_of(o[4 * j + k], mul8, (float)mul8) << (8 * k);
				__builtin_memcpy(out8 + off + 4 * j, &w4, 4);
			}
		} else {
#pragma unroll
			for (int j = 0; j < NX / 4; j++) { float4 v; v.x = o[4 * j]; v.y = o[4 * j + 1]; v.z = o[4 * j + 2]; v.w = o[4 * j + 3]; reinterpret_cast<float4 *>(out + off)[j] = v; }
		}
	}
}
// phases y / z: lines of N samples `stride` floats apart in the tile
template <int N, int KIND>
DSP_HD void block_line(const TinyArgs &t, float *p, int stride)
{
	float x[N], o[N];
#pragma unroll
	for (int j = 0; j < N; j++) x[j] = p[j * stride];
	tiny_dct<N, KIND>(t, x, o);
#pragma unroll
	for (int j = 0; j < N; j++) p[j * stride] = o[j];
}
template <int NX, int NY, int NZ, int KIND>
DSP_HD void block_lines_y(const BlockGeom &a, const TinyArgs &ty, float *lds, int cnt, int tid)
{
	if constexpr (NY > 1) {
		const int cols = cnt * NX;
		for (int l = tid; l < NZ * cols; l += BLOCK_THREADS) {
			const int z = l / cols, c = l - z * cols;
			block_line<NY, KIND>(ty, lds + z * NY * a.pitch + c, a.pitch);
		}
	}
}
template <int NX, int NY, int NZ, int KIND>
DSP_HD void block_lines_z(const BlockGeom &a, const TinyArgs &tz, float *lds, int cnt, int tid)
{
	if constexpr (NZ > 1) {
		const int cols = cnt * NX;
		for (int l = tid; l < NY * cols; l += BLOCK_THREADS) {
			const int y = l / cols, c = l - y * cols;
			block_line<NZ, KIND>(tz, lds + y * a.pitch + c, NY * a.pitch);
		}
	}
}
// The middle of the roundtrip: the LAST forward axis (z, or y for 2-D blocks), motion's coefficient filter (motion.c:683-744,
// positions are the block's own coordinates) and the same axis of the inverse on one line in registers -- one LDS round trip
// instead of three.  ZAXIS: lines along z (stride NY * pitch), else along y (stride pitch).
template <int NX, int NY, int NZ, bool ZAXIS>
DSP_HD void block_lines_mid(const BlockGeom &a, const TinyArgs &tf, const TinyArgs &ti, const MotionFilter &f, float *lds, int cnt, int tid, unsigned long long &coded)
{
	constexpr int N = ZAXIS ? NZ : NY, OUTER = ZAXIS ? NY : NZ;
	const int cols = cnt * NX;
	const int stride = ZAXIS ? NY * a.pitch : a.pitch;
	for (int l = tid; l < OUTER * cols; l += BLOCK_THREADS) {
		const int o = l / cols, c = l - o * cols;
		float *p = lds + (ZAXIS ? o * a.pitch : o * NY * a.pitch) + c;
		float x[N], y[N];
#pragma unroll
		for (int j = 0; j < N; j++) x[j] = p[j * stride];
		tiny_dct<N, KIND_REDFT10>(tf, x, y);
		if (f.enabled) {
			const int bx = c % NX;
#pragma unroll
			for (int j = 0; j < N; j++) {
				const int bz = ZAXIS ? j : o, by = ZAXIS ? o : j;
				if (bx < f.aw && by < f.ah && bz < f.ad) y[j] = motion_filter_at(f, bz, by, bx, y[j], coded);
			}
		}
		tiny_dct<N, KIND_REDFT01>(ti, y, x);
#pragma unroll
		for (int j = 0; j < N; j++) p[j * stride] = x[j];
	}
}
// rows of the tile back to global memory (four consecutive samples per lane; block extents are multiples of 4)
template <int NX, int NY, int NZ>
DSP_HD void block_store_rows(const BlockGeom &a, float *out, const float *lds, long long bout, int cnt, int tid)
{
	constexpr int QX = NX / 4;
	const int quads = NZ * NY * cnt * QX;
	for (int e = tid; e < quads; e += BLOCK_THREADS) {
		const int l = e / QX, qx = e - l * QX;
		int row, g;
		block_line_of(a, l, NZ * NY, cnt, row, g);
		const int z = row / NY, y = row - z * NY;
		*reinterpret_cast<float4 *>(out + bout + (long long)g * a.sxb_out + (long long)z * a.sz_out + (long long)y * a.sy_out + 4 * qx) =
			*reinterpret_cast<const float4 *>(lds + row * a.pitch + g * NX + 4 * qx);
	}
}
// block extents with a fused kernel: X(NX, NY, NZ) -- 4, 8 or 16 samples a side (2-D blocks, NZ = 1: up to 32)
#define DSPFFT_BLOCK_SHAPES(X) \
	X(32, 32, 1) X(32, 16, 1) X(16, 32, 1) X(32, 8, 1) X(8, 32, 1) \
	X(4, 4, 1) X(8, 4, 1) X(16, 4, 1) X(4, 8, 1) X(8, 8, 1) X(16, 8, 1) X(4, 16, 1) X(8, 16, 1) X(16, 16, 1) \
	X(4, 4, 4) X(8, 4, 4) X(16, 4, 4) X(4, 8, 4) X(8, 8, 4) X(16, 8, 4) X(4, 16, 4) X(8, 16, 4) X(16, 16, 4) \
	X(4, 4, 8) X(8, 4, 8) X(16, 4, 8) X(4, 8, 8) X(8, 8, 8) X(16, 8, 8) X(4, 16, 8) X(8, 16, 8) X(16, 16, 8) \
	X(4, 4, 16) X(8, 4, 16) X(16, 4, 16) X(4, 8, 16) X(8, 8, 16) X(16, 8, 16) X(4, 16, 16) X(8, 16, 16) X(16, 16, 16)

}  // namespace dspfft
