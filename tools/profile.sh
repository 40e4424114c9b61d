#!/bin/bash
# profile.sh TAG -- on the MI355X box: kernel-trace stats and HBM-traffic counters of the bench
# workloads; summaries land in gpurun_out/prof_TAG/ (copy the ones to keep into profiles/).
#   1. rocprofv3 --kernel-trace --stats  of  bench.py (NTT NAND, FFT NAND)
#   2. rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (one counter per pass, as
#      MI355X_MICROARCH.md prescribes) of  bench.py --steps 2 --warmup 1
TAG=${1:-run}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for TR in NTT FFT; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$TR" -- \
        python "$ROOT/bench.py" --steps 10 --warmup 2 --transform $TR --no-extra --no-cpu-baseline \
        > "$OUT/bench_$TR.json" 2> "$OUT/stats_$TR.log"
    for C in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_${TR}_$C" -- \
            python "$ROOT/bench.py" --steps 2 --warmup 1 --transform $TR --no-extra --no-cpu-baseline \
            > /dev/null 2> "$OUT/pmc_${TR}_$C.log"
    done
done
python "$ROOT/tools/profile_summary.py" "$OUT"
