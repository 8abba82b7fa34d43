// spec_inst_rt_b.hip -- explicit instantiations of one group of specialised kernels (see spec_kernels.h)
#include "spec_kernels.h"

namespace dspfft {
#define DSP_INST_RT(N, K, T, ...) \
	template int launch_col_roundtrip<ColSpec<N, K, T, __VA_ARGS__>>(const PassArgs &, const PassArgs &, const MotionFilter &, unsigned long long *, int, void *);
DSPFFT_COL_SPECS_B(DSP_INST_RT)
}  // namespace dspfft
