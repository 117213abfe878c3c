set -e
mkdir -p gpurun_out
IB="--fixture intra_main_1920x1080 --seeds 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --check-two-streams-only"
run() { name=$1; shift; "$@" > gpurun_out/x_$name.json 2> gpurun_out/x_$name.err; python3 -c "
import json,sys
d=json.load(open('gpurun_out/x_$name.json')); print('$name', d['value'], d['kernels']['k_intra'], d.get('fewer_streams'))"; }
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1
run new timeout -k 10 400 python bench.py --no-cpu-baseline --no-handover --steps 3
run new_intra timeout -k 10 200 python bench.py $IB
run new_he10 timeout -k 10 300 python bench.py --fixture randomaccess_he10_1920x1080 --seeds 1 --no-extras --no-cpu-baseline --steps 3
echo ALLDONE
