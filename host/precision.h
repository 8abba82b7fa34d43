/* host/precision.h -- the two definitions of the reference's include/precision.h the harnesses need
 * (coeff typedef :103 and the fftw(call) prefix macro :115), fixed at COEFF_PRECISION=F. */
#ifndef HOST_PRECISION_H
#define HOST_PRECISION_H
typedef float coeff;
typedef double intermediate;
#define fftw(call) fftwf_##call
#endif
