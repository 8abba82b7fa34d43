# Round-6 profiles on the GPU box (run from the repo root: bash tools/prof_round6.sh [headline|paths|motion|8k|zoom|all]); everything lands under
# gpurun_out/r06/, the summaries are copied to profiles/ by hand.  rocprofv3 runs `python3 ...` directly (never through env / bash -c); --pmc
# passes are separate runs with no tracing beside them.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06
WHAT=${1:-all}
mkdir -p $O
cd /tmp
trace() {   # name, command...
  local n=$1; shift
  rm -rf $O/$n; mkdir -p $O/$n
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o t -- "$@" > $O/$n.log 2>&1 || echo "trace $n failed"
  find $O/$n -name "*kernel_trace.csv" -delete
  python3 $R/tools/summarise_prof.py stats $O/$n $O/${n}_kernel_stats.csv
}
pmc() {     # name, command...
  local n=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${n}_$c; mkdir -p $O/${n}_$c
    rocprofv3 --pmc $c --output-format csv -d $O/${n}_$c -o p -- "$@" > $O/${n}_$c.log 2>&1 || echo "pmc $n $c failed"
  done
  python3 $R/tools/summarise_prof.py pmc $O/${n}_FETCH_SIZE $O/${n}_WRITE_SIZE $O/${n}_traffic.json
  rm -rf $O/${n}_FETCH_SIZE $O/${n}_WRITE_SIZE
}
if [ $WHAT = headline ] || [ $WHAT = all ]; then
  python3 $R/bench.py > $O/bench.json 2> $O/bench.err || true
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err || true
  python3 $R/bench.py --streams 1 --no-motion --no-scan > $O/bench_streams1.json 2> $O/bench_streams1.err || true
  trace headline python3 $R/bench.py --no-cpu-baseline --no-motion --no-scan --no-single-stream --no-fftw-abi
  trace headline_streams1 python3 $R/bench.py --no-cpu-baseline --no-motion --no-scan --no-fftw-abi --streams 1
  cd $R; bash $R/tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 || echo "pmc_traffic failed"; cp $R/gpurun_out/traffic.json $O/headline_traffic.json 2>/dev/null || true; cd /tmp
fi
if [ $WHAT = paths ] || [ $WHAT = all ]; then
  python3 $R/tools/bench_paths.py > $O/paths.json 2> $O/paths.err || true
  trace paths python3 $R/tools/bench_paths.py
  pmc paths python3 $R/tools/bench_paths.py
fi
if [ $WHAT = motion ] || [ $WHAT = all ]; then
  python3 $R/tools/bench_motion.py > $O/motion_c5.json 2> $O/motion_c5.err || true
  trace motion_c5 python3 $R/tools/prof_motion_c5.py
  cd $R; REPS=2 bash $R/tools/pmc_sq.sh r06_motion python3 tools/prof_motion_c5.py > /dev/null; cd /tmp
fi
if [ $WHAT = 8k ] || [ $WHAT = all ]; then
  trace 8k python3 $R/tools/bench_8k_quick.py
  cd $R; bash $R/tools/pmc_sq.sh r06_8k python3 tools/bench_8k_quick.py > /dev/null; cd /tmp
fi
if [ $WHAT = zoom ] || [ $WHAT = all ]; then
  trace zoom_fft python3 $R/tools/zoom_stage_probe.py 600
  trace zoom_czt python3 $R/tools/zoom_czt_probe.py 300
fi
ls $O
