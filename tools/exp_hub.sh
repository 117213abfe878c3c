#!/bin/bash
# Hub experiments on the GPU machine: N decoders behind one hub, variations by environment.
run() { echo "== N=$N L=$L $*"; env "$@" HEVCRECON_E2E_STAGES=hub timeout -k 10 300 python3 tools/e2e_only.py ${N:-16} ${L:-4} > gpurun_out/e2e_x.json 2> gpurun_out/e2e_x.err; python3 -c "
import json
d=json.load(open('gpurun_out/e2e_x.json')); h=d.get('concurrent_through_hub',{}); print(h.get('frames_per_s'), h.get('pictures_per_batch'), h.get('per_picture_ms'), d.get('error'))"; grep "hevcrecon_hub" gpurun_out/e2e_x.err | cut -c1-600; }
N=8 L=8 run A=1
N=8 L=8 run HEVCRECON_HASH_PRIORITY=normal
N=2 L=8 run A=1
