set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp
for n in headline headline_streams1; do
  rm -rf $O/$n; mkdir -p $O/$n
  if [ $n = headline ]; then extra=""; else extra="--streams 1"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o t -- python3 $R/bench.py --no-cpu-baseline --no-motion --no-scan --no-single-stream --no-fftw-abi $extra > $O/$n.log 2>&1 || echo "trace $n failed"
  find $O/$n -name "*kernel_trace.csv" -delete
  python3 $R/tools/summarise_prof.py stats $O/$n $O/${n}_kernel_stats.csv
done
tail -1 $O/headline.log | head -c 400
