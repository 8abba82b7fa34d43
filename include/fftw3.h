/*
 * fftw3.h -- the FFTW3 API subset the dspfun tools reach through `fftw(call)`
 * (reference include/precision.h:115), served by the MI355X engine (include/dspfft.h).
 *
 * The tools include <fftw3.h> (spec/spec.h:8, zoom/zoom.c:13, scan/scan.c:12, motion/motion.c:5,
 * applybasis/draw.c:12) and link -lfftw3f [-lfftw3f_threads] (scan/Makefile:4,12, motion/Makefile:20);
 * with this header first on the include path and libdspfft_hip.so on the link line they build
 * unchanged.  Pointers are HOST memory, exactly as with FFTW: execute copies the arrays to the
 * GPU, runs the device plan and copies the result back (the device-resident path for data that
 * stays on the GPU is include/dspfft.h).
 *
 * Each declaration cites the reference call sites it replaces.  Both `fftw_r2r_kind` (spec/spec.c:63,
 * zoom/zoom.c:263, motion/motion.c:538) and `fftwf_r2r_kind` (scan/scan.c:292,359 via fftw(r2r_kind))
 * spellings are used by the tools, so both exist.  Numeric values follow FFTW 3.3's public header.
 */
#ifndef DSPFFT_FFTW3_H
#define DSPFFT_FFTW3_H
#include <stddef.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif

enum fftw_r2r_kind_do_not_use_me {
	FFTW_R2HC = 0, FFTW_HC2R = 1, FFTW_DHT = 2,
	FFTW_REDFT00 = 3, FFTW_REDFT01 = 4, FFTW_REDFT10 = 5, FFTW_REDFT11 = 6,
	FFTW_RODFT00 = 7, FFTW_RODFT01 = 8, FFTW_RODFT10 = 9, FFTW_RODFT11 = 10
};
typedef enum fftw_r2r_kind_do_not_use_me fftw_r2r_kind;
typedef fftw_r2r_kind fftwf_r2r_kind;

/* planner flags (motion/motion.c:93-103 maps its --fftw-planning-method onto these; the engine
 * ignores them and NEVER touches the arrays at plan time, scan/scan.c:354-359 relies on that) */
#define FFTW_MEASURE (0U)
#define FFTW_DESTROY_INPUT (1U << 0)
#define FFTW_UNALIGNED (1U << 1)
#define FFTW_EXHAUSTIVE (1U << 3)
#define FFTW_PRESERVE_INPUT (1U << 4)
#define FFTW_PATIENT (1U << 5)
#define FFTW_ESTIMATE (1U << 6)
#define FFTW_WISDOM_ONLY (1U << 21)

/* ---------------- single precision: COEFF_PRECISION=F (scan/Makefile:1, motion/Makefile:1) ---------------- */
typedef struct fftwf_plan_s *fftwf_plan;            /* stored by value: motion/motion.c:522-524 */

float *fftwf_alloc_real(size_t n);                  /* spec.c:59 ispec.c:80 zoom.c:258 scan.c:275,352,353 motion.c:500 draw.c:66 */
void fftwf_free(void *p);                           /* spec.c:143 scan.c:550-551,562 motion.c:824 */
fftwf_plan fftwf_plan_many_r2r(int rank, const int *n, int howmany,
                               float *in, const int *inembed, int istride, int idist,
                               float *out, const int *onembed, int ostride, int odist,
                               const fftwf_r2r_kind *kind, unsigned flags);   /* spec.c:63 ispec.c:165 zoom.c:263 scan.c:292,359 motion.c:535,549 */
fftwf_plan fftwf_plan_r2r_2d(int n0, int n1, float *in, float *out,
                             fftwf_r2r_kind kind0, fftwf_r2r_kind kind1, unsigned flags);   /* draw.c:74 */
void fftwf_execute(const fftwf_plan p);             /* spec.c:64 ispec.c:166 zoom.c:264 scan.c:293,407,447 motion.c:641,753 draw.c:75 */
void fftwf_destroy_plan(fftwf_plan p);
void fftwf_cleanup(void);                           /* zoom.c:266 scan.c:563 motion.c:832 draw.c:92 */
int fftwf_init_threads(void);                       /* scan.c:289 motion.c:485 */
void fftwf_plan_with_nthreads(int nthreads);        /* scan.c:290 motion.c:486 */
void fftwf_cleanup_threads(void);                   /* scan.c:564 motion.c:837 */
int fftwf_import_wisdom_from_filename(const char *filename);   /* motion.c:519 */
int fftwf_export_wisdom_to_filename(const char *filename);     /* motion.c:557 */

/* ---------------- double precision: COEFF_PRECISION=D (spec/Makefile, zoom/Makefile default) ----------------
 * Same engine with double buffers, double arithmetic and double tables on the device
 * (dspfft_plan_many_r2r_f64 / dspfft_execute_f64): results agree with a double FFTW to ~1e-14 of max|coeff|.
 * The common frame sizes (spec_list.h DSPFFT_*_SPECS_F64: 8K, 4K, 1080p, 720p, powers of two) run compile-time-specialised double
 * kernels (3840x2160 RGB roundtrip 0.33 ms = 30 % of the 96 B/pixel roofline, 7680x4320 1.94 ms = 20.5 %, round 4); other sizes run
 * the runtime-geometry kernels. */
typedef struct fftw_plan_s *fftw_plan;
double *fftw_alloc_real(size_t n);
void fftw_free(void *p);
fftw_plan fftw_plan_many_r2r(int rank, const int *n, int howmany,
                             double *in, const int *inembed, int istride, int idist,
                             double *out, const int *onembed, int ostride, int odist,
                             const fftw_r2r_kind *kind, unsigned flags);
fftw_plan fftw_plan_r2r_2d(int n0, int n1, double *in, double *out, fftw_r2r_kind kind0, fftw_r2r_kind kind1, unsigned flags);
void fftw_execute(const fftw_plan p);
void fftw_destroy_plan(fftw_plan p);
void fftw_cleanup(void);
int fftw_init_threads(void);
void fftw_plan_with_nthreads(int nthreads);
void fftw_cleanup_threads(void);
int fftw_import_wisdom_from_filename(const char *filename);
int fftw_export_wisdom_to_filename(const char *filename);

/* ---------------- long double: COEFF_PRECISION=L (include/precision.h:73-79) ----------------
 * The GPU has no long double.  SURVEY.md 8b allows a CPU-only implementation for this precision: these entry points run a
 * plain host path (dspfun_amd/csrc/fftwl_cpu.cpp: recursive mixed-radix FFT in long double, single thread, any length) on ordinary
 * host memory.  Same call-site contract as above (plan captures the pointers and never touches the arrays; execute is repeatable). */
typedef fftw_r2r_kind fftwl_r2r_kind;
typedef struct fftwl_plan_s *fftwl_plan;
long double *fftwl_alloc_real(size_t n);
void fftwl_free(void *p);
fftwl_plan fftwl_plan_many_r2r(int rank, const int *n, int howmany,
                               long double *in, const int *inembed, int istride, int idist,
                               long double *out, const int *onembed, int ostride, int odist,
                               const fftwl_r2r_kind *kind, unsigned flags);
fftwl_plan fftwl_plan_r2r_2d(int n0, int n1, long double *in, long double *out, fftwl_r2r_kind kind0, fftwl_r2r_kind kind1, unsigned flags);
void fftwl_execute(const fftwl_plan p);
void fftwl_destroy_plan(fftwl_plan p);
void fftwl_cleanup(void);
int fftwl_init_threads(void);
void fftwl_plan_with_nthreads(int nthreads);
void fftwl_cleanup_threads(void);
int fftwl_import_wisdom_from_filename(const char *filename);
int fftwl_export_wisdom_to_filename(const char *filename);

#ifdef __cplusplus
}
#endif
#endif
