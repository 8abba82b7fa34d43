# same-box A/B of zoom's config-3 frame (fast-transform path): working tree's library vs tools/oldlib (tools/ab_oldlib.sh)
for i in 1 2 3; do
  echo "new: $(python3 tools/zoom_frame_time.py 2>/dev/null | tail -1)"
  echo "old: $(DSPFFT_LIB_PATH=$PWD/tools/oldlib/libdspfft_hip.so python3 tools/zoom_frame_time.py 2>/dev/null | tail -1)"
done
