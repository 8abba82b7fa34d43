python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "--- 3-D luma block:"; for i in 1 2; do python3 tools/bench_motion_3d.py 2>/dev/null | tail -1 | cut -c120-420; done
echo "--- K=16 entry (DSPFFT_COL_KPREF=16):"; DSPFFT_COL_KPREF=16 DSPFFT_COL_TPREF=256 python3 tools/bench_motion_3d.py 2>/dev/null | tail -1 | cut -c120-420
echo "--- motion_c5 object:"; python3 tools/bench_motion.py 2>/dev/null | tail -1 > gpurun_out/r06_motion_c5_b.json; python3 -c "
import json;d=json.load(open('gpurun_out/r06_motion_c5_b.json'));print(d['per_frame_strong']['ms_per_clip_round']);v=d['volume_3d'];print(v['ms_per_clip'],v['max_abs_roundtrip_error_0_255'],v['roundtrips_behind_that_error'])"
