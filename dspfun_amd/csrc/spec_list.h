// spec_list.h -- which (length, layout) combinations get a compile-time-specialised kernel.
// Everything else runs on the generic kernels of dct_core.h.
//
// ROW entries: X(N, C, THREADS, radices of N/2 ...)   -- last radix odd => conflict-free last stage
// COL entries: X(N, K, THREADS, radices of N ...)     -- K = tile width in floats (multiple of 4)
#pragma once

#define DSPFFT_ROW_SPECS(X)            \
	X(3840, 3, 512, 12, 10, 16)        \
	X(1920, 3, 256, 12, 5, 16)         \
	X(7680, 3, 1024, 16, 15, 16)        \
	X(960, 3, 192, 2, 16, 15)          \
	X(256, 3, 192, 8, 16)              \
	X(1920, 1, 128, 8, 8, 15)          /* round 6: (4, 16, 15) left 60 of the 128 threads busy in its radix-16 stage; motion's clip (Y + U + V) 4.05 -> 3.97 ms with this and */ \
	X(960, 1, 64, 4, 8, 15)            /* (2, 16, 15) -> (4, 8, 15) for the chroma lines; (8, 10, 12), (10, 12, 8), (6, 10, 16) / (8, 4, 15), (8, 6, 10) measured beside them */ \
	X(4096, 3, 512, 8, 16, 16)         \
	X(2560, 3, 512, 8, 10, 16)         \
	X(2048, 3, 256, 4, 16, 16)         \
	X(1280, 3, 256, 4, 10, 16)         \
	X(1024, 3, 192, 4, 8, 16)          \
	X(720, 3, 128, 4, 6, 15)           \
	X(640, 3, 128, 4, 5, 16)           \
	X(512, 3, 64, 16, 16)              \
	X(1280, 1, 128, 4, 10, 16)         \
	X(5120, 3, 512, 10, 16, 16)        \
	X(3200, 3, 512, 10, 10, 16)        \
	X(2880, 3, 512, 6, 16, 15)         \
	X(1600, 3, 256, 5, 10, 16)         \
	X(1440, 3, 256, 3, 16, 15)         \
	X(800, 3, 128, 5, 5, 16)           \
	X(3840, 1, 256, 12, 10, 16)        \
	X(2560, 1, 256, 8, 10, 16)         \
	X(4096, 1, 256, 8, 16, 16)         \
	X(2048, 1, 128, 4, 16, 16)         \
	X(1024, 1, 64, 4, 8, 16)           \
	X(7680, 1, 256, 16, 15, 16)        /* planar 8K rows: scan's colour planes on separate GPUs (BASELINE config 4, dist.ChannelShardedScan) */

// (An outer radix-2 column split -- half of the tile parked in registers so 2160-row tiles could be K = 16 wide in
// the same 69 KB -- was built and measured SLOWER at every size tried: 2160 (K=16): 57/68 us vs 49/53 us; 4320 (K=8):
// 1242 vs 1226 us per 8K roundtrip; 1080 (K=32): 85 vs 66 us.  The held half costs ~40 VGPRs, i.e. a resident
// workgroup per CU, and doubles the barrier count.  Removed; see git history for `ColSplit2`.)

// two groups, so the column kernels compile in two translation units (spec_inst_col_a/b.hip)
#define DSPFFT_COL_SPECS_A(X) \
	X(2160, 8, 512, 12, 12, 15) \
	X(1080, 16, 512, 12, 10, 9) /* every stage within one round of 512 threads (8, 9, 15: 540 butterflies in the first): fused roundtrip 552 -> 460 us */ \
	X(4320, 8, 1024, 2, 12, 12, 15) \
	X(4320, 4, 512, 2, 12, 12, 15) \
	X(540, 16, 256, 12, 5, 9) \
	X(256, 32, 128, 16, 16) /* round 6: motion's z axis (config 5 as ONE 3-D block: 256 frames a frame apart).  128-byte segments and two stages of one radix-16
	                           butterfly per thread: luma clip u8 -> u8 5.73 -> 4.55 ms, forward + inverse 5.40 -> 4.83 ms (K = 16 / 64 threads, 16 / 128 (8, 8, 4),
	                           64 / 256, 32 / 256 (8, 8, 4), 64 / 512, 32 / 64 measured beside it: tools/bench_motion_3d.py, profiles/r06_motion_3d.txt) */ \
	X(256, 16, 256, 4, 4, 16) /* inner extents that are no multiple of 32 */ \
	X(4096, 8, 1024, 16, 16, 16) \
	X(1080, 8, 256, 12, 10, 9) /* behind the K = 16 entry: taken only when a caller asks for it (be_find_spec's `prefer`): clip slices that the Infinity Cache holds */

#define DSPFFT_COL_SPECS_B(X) \
	X(4096, 4, 512, 16, 16, 16) \
	X(2048, 8, 512, 8, 16, 16) \
	X(1440, 8, 512, 8, 12, 15) \
	X(1024, 16, 512, 4, 16, 16) \
	X(720, 16, 512, 8, 10, 9) \
	X(512, 16, 256, 4, 8, 16) \
	X(480, 16, 256, 4, 8, 15) \
	X(2880, 4, 512, 12, 16, 15) \
	X(1800, 8, 512, 10, 12, 15) \
	X(1600, 8, 512, 10, 10, 16) \
	X(1200, 16, 512, 5, 16, 15) \
	X(1152, 16, 512, 8, 9, 16) \
	X(960, 16, 512, 4, 16, 15) \
	X(900, 16, 512, 10, 10, 9) \
	X(768, 16, 256, 16, 16, 3) \
	X(600, 16, 256, 5, 8, 15)

#define DSPFFT_COL_SPECS(X) DSPFFT_COL_SPECS_A(X) DSPFFT_COL_SPECS_B(X)

// Column passes split by an outer radix 2 (ColHalfSpec): X(N, K, THREADS, radices of N/2 ...).  A plan uses them (with the
// paired row pass) when the full-length tile would have to be narrower than 16 floats; the entries marked "forced only"
// exist for the CPU/GPU tests of the mechanism on small frames (DSPFFT_FORCE_SPLIT=1).
#define DSPFFT_COL_HALF_SPECS(X) \
	X(2160, 16, 512, 12, 10, 9) \
	X(4320, 16, 1024, 12, 12, 15) /* an 8K frame lives in HBM, where 64-byte row segments beat two workgroups per CU: 882 vs 922 us per roundtrip */ \
	X(4320, 8, 512, 12, 12, 15) \
	X(1080, 16, 256, 4, 9, 15)   /* forced only */ \
	X(512, 16, 256, 4, 4, 16)    /* forced only */

// row specs (N, C) that also get the paired kernel
#define DSPFFT_ROW_PAIR_SPECS(X) \
	X(7680, 1, 512, 16, 15, 16) /* 512 threads although the plain planar row runs on 256: two waiting lines of 30 samples per thread spilled (260 B per lane at 96 VGPRs, 187 us per 8K plane; 4 waves per SIMD here: 108 VGPRs, plane inverse 200 -> 143 us, fused scan step 301 -> 213) */ \
	X(3840, 3, 512, 12, 10, 16) \
	X(7680, 3, 1024, 16, 15, 16) \
	X(1920, 3, 256, 12, 5, 16) \
	X(512, 3, 64, 16, 16)

// ---- double precision (the fftw_ API: spec and zoom's default COEFF_PRECISION=D build) ----
// Same structures over double samples: X(N, C | K, THREADS, radices ...).  A slot is 16 bytes, so a 3840 x 3 line needs 92 KB of
// LDS (one workgroup per CU) and column tiles are K = 4 doubles wide = the same 32-B row segments / 69 KB as the float K = 8 tile.
#define DSPFFT_ROW_SPECS_F64(X)      \
	X(3840, 3, 512, 12, 10, 16)      \
	X(1920, 3, 512, 4, 15, 16)       \
	X(960, 3, 256, 2, 16, 15)        \
	X(3840, 1, 256, 12, 10, 16)      \
	X(1920, 1, 128, 4, 16, 15)       \
	X(512, 3, 128, 16, 16)           \
	X(4096, 3, 512, 8, 16, 16)       \
	X(2560, 3, 512, 8, 10, 16)       \
	X(2048, 3, 512, 4, 16, 16)       \
	X(1280, 3, 256, 4, 10, 16)       \
	X(1024, 3, 256, 4, 8, 16)        \
	X(1280, 1, 128, 4, 10, 16)       \
	X(7680, 3, 1024, 16, 15, 16)     /* 184 KB a line: never launched as one -- runs as channel lines only (below); 8K frames of spec / zoom's default double build */

// interleaved double lines that fill a CU's LDS on their own (92 KB: ONE workgroup per CU) run as CHANNEL LINES instead: one workgroup
// per (line, channel) on a third of the LDS (dct_spec.h RowChanSpecT, chan_work): X(N, channels, THREADS, radices of N/2 ...).
// Measured (round 3, tools/rowchan.hip and tools/cbench): 3840 x 2160 x 3 row pass in place 107-112 -> 84 us with the frame in the Infinity
// Cache, 122-127 -> 101 us over frames in HBM; roundtrip of one frame 341 -> 318 us.  What it costs: the three channel lines of a line
// write a third of every cache line each and rely on meeting in one L2 -- with a SECOND stream's column pass sharing the L2s the
// roundtrip of two frames on two streams is 8 % slower (0.76 -> 0.82 ms), so such clips should run on one stream (or DSPFFT_ROW_CHAN=0).
// 4096 x 3 doubles: a DCI 4K frame (4096 x 2160, 212 MB) 21.7K -> 25.2K Mpix/s per roundtrip (+16 %); a 4096 x 4096 frame (402 MB, HBM)
// neither gains nor loses (15.5K / 15.7K; an earlier build read -3 %).
// Not listed, measured within 1 %: 2560 x 3 and 1920 x 3 doubles (61 / 46 KB lines: two or three workgroups per CU already).
// Not listed, measured slower: 7680 x 3 floats (row pass 205 -> 215-219 us
// in place over HBM-resident frames), 3840 x 3 floats (46 KB lines, three workgroups per CU already: 46-48 -> 57-61 us).
#define DSPFFT_ROW_CHAN_SPECS_F64(X) \
	X(3840, 3, 256, 12, 10, 16)      \
	X(4096, 3, 256, 8, 16, 16)       \
	X(7680, 3, 256, 16, 15, 16)      /* round 4: the only row pass an 8K double line has (61 KB a channel line, two workgroups per CU): without it the axis ran as a
	                                    runtime-geometry column pass, 4.05 ms per roundtrip = 10 % */

#define DSPFFT_COL_SPECS_F64(X)      \
	X(2160, 4, 512, 12, 12, 15)      \
	X(1080, 8, 512, 12, 10, 9)       \
	X(540, 8, 256, 12, 5, 9)         \
	X(512, 8, 256, 4, 8, 16)         \
	X(4096, 4, 1024, 16, 16, 16)     \
	X(2048, 4, 512, 8, 16, 16)       \
	X(1440, 4, 512, 8, 12, 15)       \
	X(1024, 8, 512, 4, 16, 16)       \
	X(720, 8, 256, 6, 8, 15)         \
	X(4320, 4, 1024, 2, 12, 12, 15)  /* round 4: 8K double frames (138 KB tiles of 4 doubles, one workgroup per CU) */

// ---- zoom's x stage on the duo row kernel (dct_duo.h ZoomXLeanT, RGB lines): X(M, THREADS, radices of M/2 ...) ----
// M = scaled line length; the last radix RL is odd and THREADS = (M/2) / RL (one last-stage butterfly per thread), a multiple of 64.
// Scaled lengths without an entry keep the two-transform row pass (dspfft_execute_sum2) or the column-last order (zoom_fft.hip).
#define DSPFFT_ZOOMX_SPECS(X)        \
	X(7680, 256, 16, 16, 15)         /* BASELINE config 3: 1920 x 4 */ \
	X(5760, 192, 12, 16, 15)         /* 1920 x 3, 1440 x 4 */ \
	X(3840, 128, 8, 16, 15)          /* 1920 x 2, 960 x 4, 1280 x 3 */ \
	X(2560, 256, 16, 16, 5)          /* 1280 x 2, 640 x 4 */ \
	X(1920, 64, 8, 8, 15)            /* 960 x 2, 640 x 3, 480 x 4 */ \
	X(1280, 128, 8, 16, 5)           /* 640 x 2, 320 x 4 */

// ---- chirp-z rows (dct_czt.h): circular convolutions of P points, X(P, THREADS, radices of P ...) ----
// A line of nc coefficients and nout samples runs on the smallest P >= nc + nout - 1 listed here (8 bytes a point in LDS: two workgroups per
// CU up to P = 9600).  Longer lines than the last entry keep the dense product.
#define DSPFFT_CZT_SPECS(X)          \
	X(1200, 128, 8, 10, 15)          \
	X(2400, 256, 16, 10, 15)         \
	X(3600, 256, 16, 15, 15)         \
	X(4800, 256, 8, 8, 15, 5)        \
	X(5400, 256, 8, 9, 15, 5)        /* config 3's y axis: 1080 + 4320 - 1 */ \
	X(7200, 512, 16, 10, 9, 5)       \
	X(9600, 640, 16, 15, 8, 5)       /* config 3's x axis: 1920 + 7680 - 1.  640 threads: 600 / 640 / 1200 / 1920 butterflies per stage fill them (on 512 two stages ran a second round for 88 / 128): 3.7x zoom frame 0.695 -> 0.676 ms */ \
	X(12000, 512, 16, 10, 15, 5)     \
	X(14400, 512, 16, 15, 12, 5)     \
	X(19200, 1024, 16, 16, 15, 5)
