run() { python3 bench.py --no-cpu-baseline --no-motion --no-scan --no-fftw-abi 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1', d['value'], d['ms_per_step'], 'single', d.get('single_stream_value'), 'lat', d['frame_latency_ms']['median'])
"; }
for i in 1 2; do
unset DSPFFT_ROW_PREF DSPFFT_COL_PREF; run "default                      "
export DSPFFT_ROW_PREF=3840:1; run "row 384 (8,15,16)            "
export DSPFFT_ROW_PREF=3840:2; run "row 384 (16,15,8)            "
export DSPFFT_ROW_PREF=3840:3; run "row 512 (12,16,10)           "
unset DSPFFT_ROW_PREF
export DSPFFT_COL_PREF=2160:1; run "col 384 (12,12,15)           "
export DSPFFT_COL_PREF=2160:2; run "col 384 (15,12,12)           "
export DSPFFT_COL_PREF=2160:3; run "col 512 (9,16,15)            "
done
