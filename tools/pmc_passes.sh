#!/bin/bash
# PMC counter passes over a short single-context run of the benchmark, one rocprofv3 run per pass (the counters of a
# pass must fit the hardware: at most 8 SQ and 4 TCC counters, FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).
# usage (on the GPU machine, from the repository root): tools/pmc_passes.sh <tag> [bench args, e.g. --fixture intra_main_1920x1080 --seeds 1]
# leaves gpurun_out/pmc_<tag>/p1..p4 and summary.txt; tools/pmc_traffic.py turns them into the entry of profiles/hbm_traffic.json
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:?tag}
shift
O=$R/gpurun_out/pmc_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --streams 64 --contexts 1 --no-cpu-baseline --no-extras --check-two-streams-only $*"
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
echo "pass 1 done"
timeout -s KILL 170 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1
echo "pass 2 done"
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1
echo "pass 3 done"
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/p4 -- $B > $O/p4.log 2>&1
echo "pass 4 done"
cd $R
python3 tools/pmc_summary.py $O > $O/summary.txt
