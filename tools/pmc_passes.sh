#!/bin/bash
# PMC counter passes over the benchmark, one rocprofv3 run per pass (counters of a pass must fit the
# hardware: at most 8 SQ and 4 TCC counters, FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).
# usage: tools/pmc_passes.sh [streams]   (run on the GPU machine from the repository root)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-8}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --streams $S --no-cpu-baseline --no-extras --check-two-streams-only"
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc/p1 -- $B > $R/gpurun_out/pmc/p1.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc/p2 -- $B > $R/gpurun_out/pmc/p2.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $R/gpurun_out/pmc/p3 -- $B > $R/gpurun_out/pmc/p3.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/pmc/p4 -- $B > $R/gpurun_out/pmc/p4.log 2>&1
