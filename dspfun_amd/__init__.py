"""dspfun_amd -- MI355X-native real-even DCT engine for the dspfun tools' hot path.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/dspfft.h and include/fftw3.h)
and thin host-side mirrors of the reference call sites.
"""
from .engine import Plan, DspfftError, REDFT01, REDFT10, set_plan_effort  # noqa: F401
