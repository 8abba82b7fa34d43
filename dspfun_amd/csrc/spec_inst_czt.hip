// spec_inst_czt.hip -- explicit instantiations of the chirp-z row kernels (dct_czt.h; see spec_kernels.h)
#include "spec_kernels.h"

namespace dspfft {
#define DSP_INST_CZT(P, T, ...) \
	template int launch_czt_rows<CztSpecT<P, T, __VA_ARGS__>>(const CztArgs &, void *); \
	template int launch_czt_spectrum<CztSpecT<P, T, __VA_ARGS__>>(const CztArgs &, cf *, void *);
DSPFFT_CZT_SPECS(DSP_INST_CZT)
}  // namespace dspfft
