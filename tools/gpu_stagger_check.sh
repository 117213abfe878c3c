set -e
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py --stagger --contexts 4 --no-cpu-baseline --no-extras --steps 2 > gpurun_out/bench_stagger4.json 2> gpurun_out/bench_stagger4.err
timeout -k 10 400 python3 bench.py --stagger --contexts 2 --streams 24 --no-cpu-baseline --no-extras --steps 2 --fixture randomaccess_he10_1920x1080 --seeds 1 > gpurun_out/bench_stagger_he10.json 2> gpurun_out/bench_stagger_he10.err
timeout -k 10 400 python3 bench.py --stagger --contexts 3 --streams 48 --no-cpu-baseline --no-extras --steps 2 --fixture intra_main_1920x1080 --seeds 1 > gpurun_out/bench_stagger_intra.json 2> gpurun_out/bench_stagger_intra.err
echo ALLDONE
