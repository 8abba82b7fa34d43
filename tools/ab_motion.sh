for i in 1 2 3; do
  echo "new: $(python3 tools/prof_motion_c5.py 2>/dev/null | tail -1 | grep -o "'ms_per_clip_round': [0-9.]*")"
  echo "old: $(DSPFFT_LIB_PATH=$PWD/tools/oldlib/libdspfft_hip.so python3 tools/prof_motion_c5.py 2>/dev/null | tail -1 | grep -o "'ms_per_clip_round': [0-9.]*")"
done
