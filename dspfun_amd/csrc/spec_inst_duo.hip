// spec_inst_duo.hip -- explicit instantiations of the duo row kernels (dct_duo.h; see spec_kernels.h)
#include "spec_kernels.h"

namespace dspfft {
#define DSP_INST_ZOOMX(M, T, ...) template int launch_zoomx<RowDuoT<M, T, __VA_ARGS__>, 3>(const ZoomXArgs &, int, bool, void *);
DSPFFT_ZOOMX_SPECS(DSP_INST_ZOOMX)
}  // namespace dspfft
