#!/bin/bash
# Collects a round's evidence on the GPU machine (run from the repository root through gpurun, AFTER the counter
# passes of tools/pmc_passes.sh have been turned into profiles/hbm_traffic.json):
#   gpurun_out/final/kernel_stats.csv                rocprofv3 --kernel-trace --stats of the default bench.py command (two contexts overlapping)
#   gpurun_out/final/kernel_stats_one_context.csv    the same with one context of 64 streams (launches that share the GPU with nothing)
#   gpurun_out/final/kernel_stats_{he10,intra}.csv   the other 1080p configurations, one context
#   gpurun_out/final/bench.json, bench_he10.json, bench_intra.json   the bench.py lines (run without the profiler)
#   gpurun_out/final/h2d_ceiling.json                what the host link carries (tools/ubench/h2d_ceiling.hip)
# Copy them into profiles/ afterwards.  Every step has a time limit of its own; the script stops at the first failure.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-extras --check-two-streams-only"
trace() {  # name, bench args...
  local name=$1
  shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 $R/bench.py $Q "$@" > $O/profiled_$name.log 2>&1
  cp $O/trace_$name/*/*kernel_stats.csv $O/kernel_stats_$name.csv
  rm -rf $O/trace_$name
  echo "trace $name done"
}
trace default
mv $O/kernel_stats_default.csv $O/kernel_stats.csv
trace one_context --streams 64 --contexts 1
trace he10 --streams 64 --contexts 1 --fixture randomaccess_he10_1920x1080 --seeds 1
trace intra --streams 64 --contexts 1 --fixture intra_main_1920x1080 --seeds 1
cd $R
timeout -k 10 120 tools/ubench/bin/h2d_ceiling > $O/h2d_ceiling.json 2> $O/h2d_ceiling.err
timeout -k 10 900 python3 -u bench.py > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log > $O/bench.json
echo "bench done"
# the other 1080p configurations of the reference at the same defaults (one clip each)
timeout -k 10 600 python3 -u bench.py --fixture randomaccess_he10_1920x1080 --seeds 1 --no-extras --no-cpu-baseline > $O/bench_he10.log 2> $O/bench_he10.err
tail -1 $O/bench_he10.log > $O/bench_he10.json
timeout -k 10 600 python3 -u bench.py --fixture intra_main_1920x1080 --seeds 1 --no-extras --no-cpu-baseline > $O/bench_intra.log 2> $O/bench_intra.err
tail -1 $O/bench_intra.log > $O/bench_intra.json
echo ALLDONE
