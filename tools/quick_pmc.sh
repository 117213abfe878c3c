#!/bin/bash
# One PMC pass (instruction and LDS counters) over a short single-context run of the benchmark, summarised per kernel.
# usage (on the GPU machine, from the repository root): tools/quick_pmc.sh <out dir> [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/$1
shift
mkdir -p $O/p1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/p1 -- python3 $R/bench.py --steps 1 --warmup 0 --streams 64 --contexts 1 --no-cpu-baseline --no-extras --check-two-streams-only "$@" > $O/p1.log 2>&1
cd $R
python3 tools/pmc_summary.py $O > $O/summary.txt
rm -rf $O/p1
