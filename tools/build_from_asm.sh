#!/bin/bash
# build_from_asm.sh NAME kernels_device.s -- variant library whose DEVICE code for kernels.hip comes from the given
# (post-processed) assembly: assemble -> link the code object -> bundle -> host compile of kernels.hip around it.
# The host side and api.o are the regular ones.  Output: gpurun_variants/libnufhe_hip_NAME.so
set -e
NAME=$1; ASM=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LLVM=/opt/rocm/lib/llvm/bin
OBJ="$ROOT/gpurun_variants/obj_$NAME"
mkdir -p "$OBJ"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$ASM" -o "$OBJ/kernels_dev.o"
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$OBJ/kernels.hsaco" "$OBJ/kernels_dev.o"
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input="$OBJ/kernels.hsaco" -output="$OBJ/kernels.hipfb"
cd "$ROOT/nufhe_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -Wno-unused-value $@"
/opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$OBJ/kernels.hipfb" -c kernels.hip -o "$OBJ/kernels.o"
make api.o kernels_team8.o kernels_team.o > /dev/null     # the other translation units are the regular ones
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/gpurun_variants/libnufhe_hip_$NAME.so" "$OBJ/kernels.o" \
    kernels_team8.o kernels_team.o api.o
ls -la "$ROOT/gpurun_variants/libnufhe_hip_$NAME.so"
