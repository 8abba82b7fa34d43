set -e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/zpass; rm -rf $O; mkdir -p $O; cd /tmp
for v in default B; do
  if [ $v = B ]; then export DSPFFT_COL_KPREF=32 DSPFFT_COL_TPREF=128; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o t -- python3 $R/tools/bench_motion.py > $O/$v.log 2>&1 || echo "trace failed"
  find $O/$v -name "*kernel_trace.csv" -delete
  python3 $R/tools/summarise_prof.py stats $O/$v $O/${v}_kernel_stats.csv
  echo "== $v"; head -12 $O/${v}_kernel_stats.csv | cut -c1-170
done
