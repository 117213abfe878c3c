#!/bin/bash
# Experiment: engine contexts vs hardware queues (run on the GPU machine from the repository root).
set -e
O=gpurun_out/queues
mkdir -p $O
run() { name=$1; shift; timeout -k 10 240 python3 -u bench.py --no-extras --no-cpu-baseline --check-two-streams-only "$@" > $O/$name.log 2>&1; tail -1 $O/$name.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])"; }
run base_c2
run stag_c2 --stagger
run stag_c4 --stagger --contexts 4
export GPU_MAX_HW_QUEUES=8
run q8_c2
run q8_c4 --contexts 4
run q8_stag_c2 --stagger
run q8_stag_c4 --stagger --contexts 4
run q8_stag_c8 --stagger --contexts 8
export GPU_MAX_HW_QUEUES=16
run q16_stag_c8 --stagger --contexts 8
run q16_c4 --contexts 4
