// spec_inst_fold.hip -- explicit instantiations of the folded row kernels (dct_fold.h; see spec_kernels.h)
#include "spec_kernels.h"

namespace dspfft {
#define DSP_INST_FOLD(N, C, T, ...) \
	template int launch_row_fold<RowFoldT<N, C, T, __VA_ARGS__>, 0>(const PassArgs &, int, bool, unsigned *, void *); \
	template int launch_row_fold<RowFoldT<N, C, T, __VA_ARGS__>, 1>(const PassArgs &, int, bool, unsigned *, void *);
DSPFFT_ROW_FOLD_SPECS(DSP_INST_FOLD)
}  // namespace dspfft
