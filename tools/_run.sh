python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/r06_bench_c.json; python3 -c "
import json;d=json.load(open('gpurun_out/r06_bench_c.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac']);m=d['motion_c5'];print(m['per_frame_strong']['ms_per_clip_round'], m['volume_3d']['ms_per_clip']);print({k:v for k,v in d['scan_c4'].items() if k.startswith('ms_')})"
python3 tools/motion_slices_ab.py 20 2>&1 | grep -v amdgpu.ids | grep "whole clip\|library default"
