#!/bin/bash
# Builds the engine for gfx950 into hevc_amd/libhevcrecon.so (in-tree, so it travels with the repository
# snapshot to the GPU machine).  hipcc cross-compiles without a GPU present.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libhevcrecon.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
mkdir -p "$here/obj"
"$HIPCC" $FLAGS -c "$here/kernels.hip" -o "$here/obj/kernels.o" &
"$HIPCC" $FLAGS -x hip -c "$here/engine.cpp" -o "$here/obj/engine.o" &
g++ -O2 -std=c++17 -fPIC -Wall -c "$here/packer.cpp" -o "$here/obj/packer.o" &
g++ -O2 -std=c++17 -fPIC -Wall -c "$here/hub_client.cpp" -o "$here/obj/hub_client.o" &
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "$here/obj/kernels.o" "$here/obj/engine.o" "$here/obj/packer.o" "$here/obj/hub_client.o" -lrt
echo "built $out"
# the process that owns the GPU when many decoder processes share it (hub.cpp): the public C ABI only
g++ -O2 -std=c++17 -Wall "$here/hub.cpp" -o "$here/../hevcrecon_hub" -L"$here/.." -lhevcrecon -Wl,-rpath,'$ORIGIN' -lrt
echo "built $here/../hevcrecon_hub"
