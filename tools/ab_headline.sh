# same-box A/B of the headline bench line (value, ms_per_step, single-frame median latency): working tree's library vs tools/oldlib
for i in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export DSPFFT_LIB_PATH=$PWD/tools/oldlib/libdspfft_hip.so; else unset DSPFFT_LIB_PATH; fi
    python3 bench.py --no-scan --no-cpu-baseline --no-fftw-abi --no-motion 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['value'], d['ms_per_step'], d['frame_latency_ms']['median'])"
  done
done
