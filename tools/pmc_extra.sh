#!/bin/bash
# pmc_extra.sh TAG "GROUP1" "GROUP2" ... -- ad-hoc PMC passes over the NTT NAND bench (2 steps), one
# rocprofv3 run per quoted counter group; prints the per-launch averages of k_bootstrap and writes
# gpurun_out/pmcx_TAG/summary.json.  With no groups: lists the counters whose names mention the
# instruction cache / fetch / wait states.
TAG=${1:-x}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmcx_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then
    rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|SQC_|SQ_WAIT|SQ_INST_LEVEL|SQ_LEVEL|SQ_IFETCH|DCACHE" \
        | tee "$OUT/list.txt" | head -120
    exit 0
fi
i=0
for G in "$@"; do
    rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/pmc_NTT_g$i" -- \
        python "$ROOT/bench.py" --steps 2 --warmup 1 --transform ${TRANSFORM:-NTT} --no-extra --no-cpu-baseline \
        > /dev/null 2> "$OUT/pmc_g$i.log"
    i=$((i+1))
done
python - "$OUT" <<'EOF'
import glob, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(sys.argv[1]), '..', 'tools'))
import profile_summary as ps
out = sys.argv[1]
res = {}
kern = 'k_bootstrap_fft' if os.environ.get('TRANSFORM') == 'FFT' else 'k_bootstrap'
for d in sorted(glob.glob(os.path.join(out, 'pmc_NTT_g*'))):
    if os.path.isdir(d):
        vals, dur = ps.pass_counters(d, kern)
        for c, v in vals.items():
            res[c] = {'per_launch': v, 'kernel_ns': dur}
            print('%-36s %18.1f   (kernel %.3f ms)' % (c, v, (dur or 0) * 1e-6))
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1, sort_keys=True)
EOF
