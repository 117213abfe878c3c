#!/bin/bash
# Collects the round's evidence on the GPU machine (run from the repository root through gpurun):
#   gpurun_out/final/kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench.py command
#   gpurun_out/final/pmc_summary.txt    per-kernel averages of the PMC passes (tools/pmc_passes.sh)
#   gpurun_out/final/hbm_traffic.json   HBM bytes per launch from FETCH_SIZE / WRITE_SIZE
#   gpurun_out/final/bench.json         the bench.py line (run without the profiler)
# Copy them into profiles/ afterwards.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --no-extras --check-two-streams-only > $O/bench_profiled.log 2>&1
# the same with one context (64 streams): launches that do not share the GPU with another context's kernels
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -- python3 $R/bench.py --no-cpu-baseline --no-extras --check-two-streams-only --streams 64 --contexts 1 > $O/bench_profiled_one_context.log 2>&1
cp $O/trace1/*/*kernel_stats.csv $O/kernel_stats_one_context.csv
rm -rf $O/trace1
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv
rm -rf $O/trace
cd $R
bash tools/pmc_passes.sh 128   # bench defaults: 128 streams over 2 contexts = 64 pictures per launch
python3 tools/pmc_summary.py gpurun_out/pmc > $O/pmc_summary.txt
python3 tools/pmc_traffic.py gpurun_out/pmc 64 randomaccess_main_1920x1080 8 > $O/hbm_traffic.json
rm -rf gpurun_out/pmc/p*/
cp $O/hbm_traffic.json profiles/hbm_traffic.json  # the bench line below reads its roofline.traffic from there
timeout -k 10 900 python3 -u bench.py > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/bench.json
# the other 1080p configurations of the reference at the same defaults (one clip each)
timeout -k 10 600 python3 -u bench.py --fixture randomaccess_he10_1920x1080 --seeds 1 --no-extras --no-cpu-baseline > $O/bench_he10.log 2>&1
tail -1 $O/bench_he10.log > $O/bench_he10.json
timeout -k 10 600 python3 -u bench.py --fixture intra_main_1920x1080 --seeds 1 --no-extras --no-cpu-baseline > $O/bench_intra.log 2>&1
tail -1 $O/bench_intra.log > $O/bench_intra.json
