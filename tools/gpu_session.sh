set -e
mkdir -p gpurun_out
IB="--fixture intra_main_1920x1080 --seeds 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --check-two-streams-only"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1
HEVCRECON_LIB=$PWD/hevc_amd/libhevcrecon_stamps.so timeout -k 10 200 python bench.py $IB --contexts 1 --streams 64 > gpurun_out/bench_intra_stamps.json 2> gpurun_out/bench_intra_stamps.err
timeout -k 10 200 python bench.py $IB > gpurun_out/bench_intra.json 2> gpurun_out/bench_intra.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-handover > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo ALLDONE
