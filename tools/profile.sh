#!/bin/bash
# profile.sh TAG -- on the MI355X box: kernel-trace stats and PMC counters of the bench workloads;
# summaries land in gpurun_out/prof_TAG/ (copy the ones to keep into profiles/).
#   1. rocprofv3 --kernel-trace --stats  of  bench.py --steps 10 --warmup 2 (NTT NAND, FFT NAND)
#   2. rocprofv3 --kernel-trace --pmc <one group per pass> of  bench.py --steps 2 --warmup 1:
#      SQ issue/wait counters, GRBM_GUI_ACTIVE, FETCH_SIZE and WRITE_SIZE in their own passes
#      (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
# -> kernel_stats_<TR>.csv, pmc_<TR>.json (per-launch averages of the bootstrap kernel), pmc_traffic.json
TAG=${1:-run}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
GROUPS_PMC=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
  "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM"
  "FETCH_SIZE"
  "WRITE_SIZE"
)
for TR in ${TRANSFORMS:-NTT FFT XFFT}; do
    ARGS="--transform $TR"
    [ "$TR" = "XFFT" ] && ARGS="--transform NTT --engine exact-fft"      # the exact fp64 engine of the NTT path
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$TR" -- \
        python "$ROOT/bench.py" --steps 10 --warmup 2 $ARGS --no-extra --no-cpu-baseline \
        > "$OUT/bench_$TR.json" 2> "$OUT/stats_$TR.log"
    i=0
    for G in "${GROUPS_PMC[@]}"; do
        rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/pmc_${TR}_g$i" -- \
            python "$ROOT/bench.py" --steps 2 --warmup 1 $ARGS --no-extra --no-cpu-baseline \
            > /dev/null 2> "$OUT/pmc_${TR}_g$i.log"
        i=$((i+1))
    done
done
python "$ROOT/tools/profile_summary.py" "$OUT"
