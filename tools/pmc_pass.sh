#!/bin/bash
# pmc_pass.sh TAG "COUNTER1 COUNTER2 ..." -- one rocprofv3 --pmc pass (kernel-trace only) of a short NTT bench
# (BENCH_ARGS="--bits 256 --engine exact-fft" for another workload)
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $@ --output-format csv -d "$OUT" -- \
    python "$ROOT/bench.py" --steps 2 --warmup 1 --no-extra --no-cpu-baseline $BENCH_ARGS > /dev/null 2> "$OUT/log.txt"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
path = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in path:
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name']
        if 'k_bootstrap' not in k: continue
        acc[r['Counter_Name']][r['Dispatch_Id']].append(float(r['Counter_Value']))
for c, d in acc.items():
    vals = [sum(v) for v in d.values()]
    print(c, sum(vals) / len(vals))
PY
