run() { python3 tools/bench_motion.py 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 per_frame', d['per_frame_strong']['ms_per_clip_round'])
"; }
for i in 1 2; do
for k in 0 1 2 3 4; do export DSPFFT_COL_KPREF_SKIP=$k; run "1080 K=8 entry $k"; done
done
