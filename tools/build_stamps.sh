#!/bin/bash
# Diagnostic build of the engine with per-phase cycle stamps in the intra kernel (hevc_amd/libhevcrecon_stamps.so;
# use it with HEVCRECON_LIB=<path>: the totals are printed when a context is destroyed).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/../hevc_amd/csrc" && pwd)"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHEVCRECON_INTRA_STAMPS -Wno-unused-value -Wno-unused-result"
mkdir -p "$here/obj_st"
/opt/rocm/bin/hipcc $F -c "$here/kernels.hip" -o "$here/obj_st/kernels.o" &
/opt/rocm/bin/hipcc $F -x hip -c "$here/engine.cpp" -o "$here/obj_st/engine.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/../libhevcrecon_stamps.so" "$here/obj_st/kernels.o" "$here/obj_st/engine.o" "$here/obj/packer.o"
