run() { python3 tools/bench_motion.py 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 per_frame', d['per_frame_strong']['ms_per_clip_round'], 'volume', d['volume_3d']['ms_per_clip'])
"; }
for i in 1 2; do
unset DSPFFT_ROW_PREF; run "default"
export DSPFFT_ROW_PREF=1920:1; run "1920 (8,8,15)          "
export DSPFFT_ROW_PREF=1920:2; run "1920 (8,10,12)         "
export DSPFFT_ROW_PREF=1920:3; run "1920 (10,12,8)         "
export DSPFFT_ROW_PREF=1920:4; run "1920 (6,10,16)         "
export DSPFFT_ROW_PREF=960:1; run "960 (4,8,15)           "
export DSPFFT_ROW_PREF=960:2; run "960 (8,4,15)           "
export DSPFFT_ROW_PREF=960:3; run "960 (8,6,10)           "
export DSPFFT_ROW_PREF=1920:1,960:1; run "1920 (8,8,15) + 960 (4,8,15)"
done
