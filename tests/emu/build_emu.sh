#!/bin/bash
# Builds tests/emu/libhevcrecon_emu.so: the engine sources compiled for the CPU against hip_emu.h.
# Debug/test harness only (see hip_emu.h); never loaded by the hevc_amd package.
# usage: build_emu.sh [--asan]   (--asan: instead, tools/asan_driver.cpp with the host side and the emulated kernels in
#        one executable under AddressSanitizer + UndefinedBehaviorSanitizer: tests/emu/asan_driver, see its header)
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
src="$here/../../hevc_amd/csrc"
if [ "${1:-}" = "--asan" ]; then
  g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -fno-omit-frame-pointer \
      -DHEVCRECON_EMULATE -I"$here" -I"$src" -x c++ "$src/kernels.hip" "$src/engine.cpp" "$src/packer.cpp" "$src/hub_client.cpp" "$here/hip_emu.cpp" \
      "$here/../../tools/asan_driver.cpp" -o "$here/asan_driver" -lrt
  echo "built $here/asan_driver"
  exit 0
fi
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -DHEVCRECON_EMULATE -I"$here" -I"$src" \
    -x c++ "$src/kernels.hip" "$src/engine.cpp" "$src/packer.cpp" "$src/hub_client.cpp" "$here/hip_emu.cpp" -o "$here/libhevcrecon_emu.so" -lrt
echo "built $here/libhevcrecon_emu.so"
# the hub process (hub.cpp) against the emulated engine: the many-decoders-one-GPU-process path without a GPU
g++ -O1 -g -std=c++17 -pthread "$src/hub.cpp" -o "$here/hevcrecon_hub_emu" "$here/libhevcrecon_emu.so" -Wl,-rpath,"$here" -lrt
echo "built $here/hevcrecon_hub_emu"
