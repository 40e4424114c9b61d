#!/bin/bash
# build_variant.sh NAME "-DFLAG=..." -- an alternate build of libnufhe_hip.so for kernel-tuning experiments:
# gpurun_variants/libnufhe_hip_NAME.so (git-ignored, travels to the GPU box); select it with
# NUFHE_HIP_LIBRARY=$PWD/gpurun_variants/libnufhe_hip_NAME.so
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_variants/obj_$NAME"
cd "$ROOT/nufhe_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -Wno-unused-value $@"
/opt/rocm/bin/hipcc $FLAGS -c kernels.hip -o "$ROOT/gpurun_variants/obj_$NAME/kernels.o" -Rpass-analysis=kernel-resource-usage 2> "$ROOT/gpurun_variants/obj_$NAME/kernels.log" &
/opt/rocm/bin/hipcc $FLAGS -c api.hip -o "$ROOT/gpurun_variants/obj_$NAME/api.o" &
/opt/rocm/bin/hipcc $FLAGS ${TEAM8_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp -DFF_MULWIDE_PLAIN} -c kernels_team8.hip -o "$ROOT/gpurun_variants/obj_$NAME/kernels_team8.o" &
/opt/rocm/bin/hipcc $FLAGS ${TEAM_FLAGS--DFF_MULWIDE_PLAIN} -c kernels_team.hip -o "$ROOT/gpurun_variants/obj_$NAME/kernels_team.o" &
/opt/rocm/bin/hipcc $FLAGS -c kernels_xfft.hip -o "$ROOT/gpurun_variants/obj_$NAME/kernels_xfft.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/gpurun_variants/libnufhe_hip_$NAME.so" \
    "$ROOT/gpurun_variants/obj_$NAME/kernels.o" "$ROOT/gpurun_variants/obj_$NAME/kernels_team8.o" \
    "$ROOT/gpurun_variants/obj_$NAME/kernels_team.o" "$ROOT/gpurun_variants/obj_$NAME/kernels_xfft.o" "$ROOT/gpurun_variants/obj_$NAME/api.o"
for K in _Z11k_bootstrapILi1EEv8BrLaunch _Z15k_bootstrap_fft8BrLaunch; do
    echo -n "$K: "; grep -A12 "Function Name: $K" "$ROOT/gpurun_variants/obj_$NAME/kernels.log" | grep -E " VGPRs:|ScratchSize" | sed 's/\[-Rpass[^]]*\]//g; s/.*remark://' | tr -s ' \n' ' '; echo
done
ls -la "$ROOT/gpurun_variants/libnufhe_hip_$NAME.so"
