#!/bin/bash
# Two PMC passes over a short all-intra run (one context, 64 streams): where the intra kernel's time goes.
# usage (on the GPU machine, from the repository root): tools/pmc_intra.sh [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_intra
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --fixture intra_main_1920x1080 --seeds 1 --steps 1 --warmup 0 --streams 64 --contexts 1 --no-cpu-baseline --no-extras --check-two-streams-only"
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p1 -- $B "$@" > $O/p1.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -- $B "$@" > $O/p2.log 2>&1
cd $R
python3 tools/pmc_summary.py $O > $O/summary.txt
