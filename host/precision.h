/* host/precision.h -- the two definitions of the reference's include/precision.h the harnesses need
 * (coeff typedef :103 and the fftw(call) prefix macro :115).  COEFF_PRECISION=F unless the harness is built
 * with -DCOEFF_PRECISION_D, which selects the reference's default for spec (spec/Makefile:1, precision.h:50-53). */
#ifndef HOST_PRECISION_H
#define HOST_PRECISION_H
#ifdef COEFF_PRECISION_D
typedef double coeff;
typedef long double intermediate;
#define fftw(call) fftw_##call
#else
typedef float coeff;
typedef double intermediate;
#define fftw(call) fftwf_##call
#endif
#endif
