for i in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export DSPFFT_LIB_PATH=$PWD/tools/oldlib/libdspfft_hip.so; else unset DSPFFT_LIB_PATH; fi
  python3 tools/bench_motion.py 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v (960 REDFT10 rows at 8 waves = new, 6 = old) per_frame', d['per_frame_strong']['ms_per_clip_round'])
"
done; done
