// spec_inst_split.hip -- explicit instantiations of the outer-radix-2 column split kernels (see spec_kernels.h)
#include "spec_kernels.h"

namespace dspfft {
#define DSP_INST_HALF(N, K, T, ...) \
	template int launch_col_half<ColHalfSpec<N, K, T, __VA_ARGS__>, 0>(const PassArgs &, int, void *); \
	template int launch_col_half<ColHalfSpec<N, K, T, __VA_ARGS__>, 1>(const PassArgs &, int, void *);
#define DSP_INST_PAIR(N, C, T, ...) \
	template int launch_row_pair<RowSpec<N, C, T, __VA_ARGS__>, 0>(const PassArgs &, int, void *); \
	template int launch_row_pair<RowSpec<N, C, T, __VA_ARGS__>, 1>(const PassArgs &, int, void *);
DSPFFT_COL_HALF_SPECS(DSP_INST_HALF)
DSPFFT_ROW_PAIR_SPECS(DSP_INST_PAIR)
}  // namespace dspfft
