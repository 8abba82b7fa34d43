# host/scan_gpu (scan/scan.c's frame loop over include/fftw3.h, host buffers) on a 7680x4320 RGB float frame, step 2^20 = 32 output frames:
# what one fftw(execute) costs with the input going up as non-zero blocks (default) and dense (DSPFFT_UPLOAD_THREADS=0).  Run from the repo root on the GPU box.
set -e
R=$(pwd)
python3 - <<'PY'
import numpy as np
rng = np.random.default_rng(8)
with open("/tmp/in8k.pf", "wb") as f:
    f.write(b"PF\n7680 4320\n-1.0\n")
    rng.random((4320, 7680, 3), dtype=np.float32).tofile(f)
PY
echo "# cpu quota: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for t in default 0 4 8 12 16 24; do
  if [ $t = default ]; then unset DSPFFT_UPLOAD_THREADS; else export DSPFFT_UPLOAD_THREADS=$t; fi
  echo "## DSPFFT_UPLOAD_THREADS=$t"
  $R/host/scan_gpu /tmp/in8k.pf /tmp/out8k.pf 1048576 2>&1 | tail -2
done
rm -f /tmp/in8k.pf /tmp/out8k.pf
