// spec_inst_col_b.hip -- explicit instantiations of one group of specialised kernels (see spec_kernels.h)
#include "spec_kernels.h"

namespace dspfft {
#define DSP_INST_COL(N, K, T, ...) \
	template int launch_col_spec<ColSpec<N, K, T, __VA_ARGS__>, 0>(const PassArgs &, int, void *); \
	template int launch_col_spec<ColSpec<N, K, T, __VA_ARGS__>, 1>(const PassArgs &, int, void *);
DSPFFT_COL_SPECS_B(DSP_INST_COL)
}  // namespace dspfft
