#!/bin/bash
# What a development iteration runs on the GPU machine (through gpurun, from the repository root): the GPU test
# suite, then the default benchmark line and the all-intra one without the CPU baseline.  Every step has its own
# time limit and the script stops at the first failure.
set -e
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-handover --steps 3 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout -k 10 200 python3 bench.py --fixture intra_main_1920x1080 --seeds 1 --steps 2 --no-cpu-baseline --no-extras --check-two-streams-only > gpurun_out/bench_intra.json 2> gpurun_out/bench_intra.err
echo ALLDONE
