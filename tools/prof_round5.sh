# Round-5 profiles on the GPU box (run from the repo root: bash tools/prof_round5.sh <what>); everything lands under gpurun_out/r05/, the
# summaries are copied to profiles/ by hand.  rocprofv3 runs `python3 ...` directly (never through env / bash -c); --pmc passes are
# separate runs with no tracing beside them.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05
WHAT=${1:-motion}
mkdir -p $O
cd /tmp
trace() {   # name, command...
  local n=$1; shift
  rm -rf $O/$n; mkdir -p $O/$n
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o t -- "$@" > $O/$n.log 2>&1 || echo "trace $n failed"
  find $O/$n -name "*kernel_trace.csv" -delete
  python3 $R/tools/summarise_prof.py stats $O/$n $O/${n}_kernel_stats.csv
}
case $WHAT in
  motion) trace motion_c5 python3 $R/tools/prof_motion_c5.py; cd $R; REPS=2 bash $R/tools/pmc_sq.sh r05_motion python3 tools/prof_motion_c5.py > /dev/null ;;
  8k)     trace 8k python3 $R/tools/bench_8k_quick.py; cd $R; bash $R/tools/pmc_sq.sh r05_8k python3 tools/bench_8k_quick.py > /dev/null ;;
esac
