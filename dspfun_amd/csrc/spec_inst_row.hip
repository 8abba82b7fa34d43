// spec_inst_row.hip -- explicit instantiations of one group of specialised kernels (see spec_kernels.h)
#include "spec_kernels.h"

namespace dspfft {
// the 8-bit variants exist only for planar rows (RowSpec::U8_OK): instantiating this holder takes the launcher's address
// there and nothing otherwise
template <class S, int KIND, bool OK = S::U8_OK> struct U8Inst { static constexpr int (*fn)(const PassArgs &, const U8IO &, int, void *) = &launch_row_spec_u8<S, KIND>; };
template <class S, int KIND> struct U8Inst<S, KIND, false> { static constexpr void *fn = nullptr; };
#define DSP_INST_ROW(N, C, T, ...) \
	template int launch_row_spec<RowSpec<N, C, T, __VA_ARGS__>, 0>(const PassArgs &, int, void *); \
	template int launch_row_spec<RowSpec<N, C, T, __VA_ARGS__>, 1>(const PassArgs &, int, void *); \
	template int launch_row_sum2<RowSpec<N, C, T, __VA_ARGS__>>(const PassArgs &, const PassArgs &, int, void *); \
	template struct U8Inst<RowSpec<N, C, T, __VA_ARGS__>, 0>; \
	template struct U8Inst<RowSpec<N, C, T, __VA_ARGS__>, 1>;
DSPFFT_ROW_SPECS(DSP_INST_ROW)
}  // namespace dspfft
