#!/bin/bash
# Builds tests/emu/libhevcrecon_emu.so: the engine sources compiled for the CPU against hip_emu.h.
# Debug/test harness only (see hip_emu.h); never loaded by the hevc_amd package.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
src="$here/../../hevc_amd/csrc"
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -DHEVCRECON_EMULATE -I"$here" -I"$src" \
    -x c++ "$src/kernels.hip" "$src/engine.cpp" "$src/packer.cpp" "$src/hub_client.cpp" "$here/hip_emu.cpp" -o "$here/libhevcrecon_emu.so" -lrt
echo "built $here/libhevcrecon_emu.so"
# the hub process (hub.cpp) against the emulated engine: the many-decoders-one-GPU-process path without a GPU
g++ -O1 -g -std=c++17 -pthread "$src/hub.cpp" -o "$here/hevcrecon_hub_emu" "$here/libhevcrecon_emu.so" -Wl,-rpath,"$here" -lrt
echo "built $here/hevcrecon_hub_emu"
