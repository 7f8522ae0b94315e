#!/bin/bash
# tools/build_asm_variant.sh <name> <asm-filter> [extra hipcc -D flags...]
# A/B build of the library with a post-pass over the decode kernel's gfx950 ASSEMBLY: the device side is compiled to text,
# piped through <asm-filter> (a command reading the assembly on stdin, writing it to stdout), assembled, linked and bundled
# exactly like hipcc does it (hipcc -### shows the steps), then embedded into the host object. -> zxc_amd/libzxc_<name>.so
set -e
cd "$(dirname "$0")/../zxc_amd/csrc"
name=$1; filter=$2; shift; shift
B=build/var_$name; mkdir -p $B
LLVM=/opt/rocm/lib/llvm/bin
HIPCC=/opt/rocm/bin/hipcc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -DZXC_EXPERIMENT"
$HIPCC $F "$@" --cuda-device-only -S zxc_decode_kernel.hip -o $B/dk_dev.s
$filter < $B/dk_dev.s > $B/dk_dev_f.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $B/dk_dev_f.s -o $B/dk_dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $B/dk_dev.out $B/dk_dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$B/dk_dev.out -output=$B/dk.hipfb
$HIPCC $F "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $B/dk.hipfb -c zxc_decode_kernel.hip -o $B/dk.o
$HIPCC $F "$@" -c zxc_hip_shim.hip -o $B/shim.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o ../libzxc_$name.so $B/dk.o build/zxc_encode_kernel.o $B/shim.o build/zxc_host.o
echo built ../libzxc_$name.so
