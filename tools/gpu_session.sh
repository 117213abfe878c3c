#!/bin/bash
# What a development iteration runs on the GPU machine (through gpurun, from the repository root): the GPU test
# suite, the host-link ceiling, then the default benchmark line.  Every step has its own time limit and the script
# stops at the first failure.
set -e
mkdir -p gpurun_out
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1
timeout -k 10 120 tools/ubench/bin/h2d_ceiling > gpurun_out/h2d_ceiling.json 2> gpurun_out/h2d_ceiling.err
timeout -k 10 600 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo ALLDONE
