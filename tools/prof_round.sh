# Round-end measurement on the GPU box: bench.py in both stream modes, and the rocprofv3 kernel-trace summaries of the
# same two commands.  Run from the repo root: bash tools/prof_round.sh   (outputs under gpurun_out/)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
rm -rf $R/gpurun_out/prof2 $R/gpurun_out/prof1
mkdir -p $R/gpurun_out/prof2 $R/gpurun_out/prof1
python3 bench.py > $R/gpurun_out/bench_default.json 2> $R/gpurun_out/bench_default.err
python3 bench.py --streams 1 > $R/gpurun_out/bench_streams1.json 2> $R/gpurun_out/bench_streams1.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof2 -o s2 -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof1 -o s1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 > $R/gpurun_out/prof1.log 2>&1
find $R/gpurun_out/prof2 $R/gpurun_out/prof1 -name "*kernel_trace.csv" -delete
find $R/gpurun_out/prof2 $R/gpurun_out/prof1 -name "*kernel_stats.csv"
tail -1 $R/gpurun_out/bench_default.json | cut -c1-120
tail -1 $R/gpurun_out/bench_streams1.json | cut -c1-120
