#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc -D flags...] : A/B build of the library as zxc_amd/libzxc_<name>.so
set -e
cd "$(dirname "$0")/../zxc_amd/csrc"
name=$1; shift
mkdir -p build/var_$name
HIPCC=/opt/rocm/bin/hipcc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -DZXC_EXPERIMENT"
$HIPCC $F "$@" -c zxc_decode_kernel.hip -o build/var_$name/dk.o
$HIPCC $F "$@" -c zxc_hip_shim.hip -o build/var_$name/shim.o   # (-DZXC_EXPERIMENT exports zxc_mi355x__set_debug)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libzxc_$name.so build/var_$name/dk.o build/zxc_encode_kernel.o build/var_$name/shim.o build/zxc_host.o
echo built ../libzxc_$name.so
