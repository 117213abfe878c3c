set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --streams 8 --no-cpu-baseline"
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc/p1 -- $B > $R/gpurun_out/pmc/p1.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_LDS TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $R/gpurun_out/pmc/p2 -- $B > $R/gpurun_out/pmc/p2.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $R/gpurun_out/pmc/p3 -- $B > $R/gpurun_out/pmc/p3.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/pmc/p4 -- $B > $R/gpurun_out/pmc/p4.log 2>&1
ls -R $R/gpurun_out/pmc | head -30
