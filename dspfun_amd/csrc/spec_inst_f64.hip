// spec_inst_f64.hip -- explicit instantiations of the double-precision specialised kernels (see spec_kernels.h)
#include "spec_kernels.h"

namespace dspfft {
#define DSP_INST_ROW_D(N, C, T, ...) \
	template int launch_row_spec<RowSpecT<double, N, C, T, __VA_ARGS__>, 0>(const PassArgsD &, int, void *); \
	template int launch_row_spec<RowSpecT<double, N, C, T, __VA_ARGS__>, 1>(const PassArgsD &, int, void *);
#define DSP_INST_COL_D(N, K, T, ...) \
	template int launch_col_spec<ColSpecT<double, N, K, T, __VA_ARGS__>, 0>(const PassArgsD &, int, void *); \
	template int launch_col_spec<ColSpecT<double, N, K, T, __VA_ARGS__>, 1>(const PassArgsD &, int, void *);
DSPFFT_ROW_SPECS_F64(DSP_INST_ROW_D)
DSPFFT_COL_SPECS_F64(DSP_INST_COL_D)
}  // namespace dspfft
