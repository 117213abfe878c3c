#!/bin/bash
# Decoded frames/s of N decoder processes sharing the GPU, against the number of hardware queues per process and the
# copy engines (runs tools/e2e_only.py; on the GPU machine, from the repository root).
show() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
for k in ('concurrent','concurrent_host_hash'):
    if k in d: print(' ', k, d[k]['frames_per_s'], d[k]['per_picture_ms'])
if 'error' in d: print(d['error'])
" $1; }
mkdir -p gpurun_out
for cfg in "$@"; do
  IFS=: read n q sdma <<< "$cfg"
  echo "== procs $n hwq ${q:-default} sdma ${sdma:-default}"
  env ${q:+HEVCRECON_E2E_HWQ=$q} ${sdma:+HSA_ENABLE_SDMA=$sdma} timeout -k 10 200 python3 tools/e2e_only.py $n 4 > gpurun_out/e2e_m.json 2> gpurun_out/e2e_m.err || { tail -3 gpurun_out/e2e_m.err; exit 1; }
  show gpurun_out/e2e_m.json
done
