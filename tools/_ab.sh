for i in 1 2; do
unset DSPFFT_COL_PREF; echo "default (2,12,12,15): $(python3 tools/zoom_frame_time.py 2>/dev/null | tail -1)"
for k in 2 3 4 5; do export DSPFFT_COL_PREF=4320:$k; echo "4320:$k: $(python3 tools/zoom_frame_time.py 2>/dev/null | tail -1)"; done
done
