#!/bin/bash
# build_xfft_variant.sh NAME "-DFLAG=..." -- an alternate build of the exact-FFT unit only (kernels_xfft.hip) linked with
# the in-tree objects of everything else (run `make -C nufhe_amd/csrc` first): gpurun_variants/libnufhe_hip_NAME.so.
# Prints the resource use of k_bootstrap_xfft.  Select with NUFHE_HIP_LIBRARY, or tools/bench_variants.sh --engine exact-fft
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_variants/obj_$NAME"
cd "$ROOT/nufhe_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -Wno-unused-value $@"
/opt/rocm/bin/hipcc $FLAGS -c kernels_xfft.hip -o "$ROOT/gpurun_variants/obj_$NAME/kernels_xfft.o" -Rpass-analysis=kernel-resource-usage > "$ROOT/gpurun_variants/obj_$NAME/build.log" 2>&1 || { grep -B2 -A6 "error" "$ROOT/gpurun_variants/obj_$NAME/build.log"; exit 1; }
grep -A12 "Function Name: _Z16k_bootstrap_xfft" "$ROOT/gpurun_variants/obj_$NAME/build.log" | grep -E "VGPRs:|ScratchSize|Occupancy" | \
    sed 's/\[-Rpass[^]]*\]//g; s/kernels_xfft.hip:[0-9]*:1: remark://g' | tr -s ' \n' ' '
echo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/gpurun_variants/libnufhe_hip_$NAME.so" \
    kernels.o kernels_team8.o kernels_team.o api.o "$ROOT/gpurun_variants/obj_$NAME/kernels_xfft.o"
