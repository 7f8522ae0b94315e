#!/bin/bash
# tools/build_enc_variant.sh <name> [extra hipcc -D flags...] : A/B build of the ENCODE kernel as zxc_amd/libzxc_<name>.so
set -e
cd "$(dirname "$0")/../zxc_amd/csrc"
name=$1; shift
mkdir -p build/var_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -DZXC_EXPERIMENT"
/opt/rocm/bin/hipcc $F "$@" -c zxc_encode_kernel.hip -o build/var_$name/ek.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libzxc_$name.so build/zxc_decode_kernel.o build/var_$name/ek.o build/zxc_hip_shim.o build/zxc_host.o
echo built ../libzxc_$name.so
