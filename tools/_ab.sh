for i in 1 2; do
export DSPFFT_COL_KPREF=32 DSPFFT_COL_TPREF=128; echo "B 32/128 16,16: $(python3 tools/bench_motion_3d.py 2>/dev/null | tail -1 | cut -c150-400)"
export DSPFFT_COL_KPREF=32 DSPFFT_COL_TPREF=256; echo "E 32/256 8,8,4: $(python3 tools/bench_motion_3d.py 2>/dev/null | tail -1 | cut -c150-400)"
export DSPFFT_COL_KPREF=64 DSPFFT_COL_TPREF=512; echo "G 64/512 8,8,4: $(python3 tools/bench_motion_3d.py 2>/dev/null | tail -1 | cut -c150-400)"
export DSPFFT_COL_KPREF=32 DSPFFT_COL_TPREF=64; echo "F 32/64 16,16: $(python3 tools/bench_motion_3d.py 2>/dev/null | tail -1 | cut -c150-400)"
done
