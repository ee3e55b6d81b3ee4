#!/bin/bash
# Compile one .hip source with a sed script applied to its gfx950 ASSEMBLY (device side) before it is assembled:
#   scripts/asm_patch_obj.sh SRC.hip OUT.o 'SED-EXPR' [extra hipcc flags...]
# The steps are hipcc's own (hipcc -v): device -S, [sed], assemble, lld, clang-offload-bundler, host compile with the bundle.
set -e
SRC=$1; OUT=$2; SEDX=$3; shift; shift; shift
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fvisibility=hidden $*"
hipcc $FLAGS --cuda-device-only -S -o $T/dev.s $SRC
sed -E "$SEDX" $T/dev.s > $T/dev_p.s
$LLVM/clang -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/dev_p.s -o $T/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c $SRC -o $OUT
echo "patched lines: $(diff $T/dev.s $T/dev_p.s | grep -c '^>')" >&2
rm -rf $T
