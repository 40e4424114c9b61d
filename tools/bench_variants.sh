#!/bin/bash
# bench_variants.sh [bench args] -- NAND bench line (kernel ms) for the in-tree library and every
# gpurun_variants/libnufhe_hip_*.so (tools/build_variant.sh); run on the GPU box
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for lib in "" gpurun_variants/libnufhe_hip_*.so; do
    [ -n "$lib" ] && [ ! -f "$lib" ] && continue
    NUFHE_HIP_LIBRARY=${lib:+$ROOT/$lib} python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline "$@" 2>/dev/null | tail -1 | \
        python -c "
import json, sys
line = sys.stdin.read()
try:
    d = json.loads(line)
    print('%-46s step %.3f ms  K1 %.3f ms  correct %s' % ('${lib:-in-tree}', d['ms_per_step'], d['roofline']['kernel_ms'], d['correct']))
except ValueError:
    print('%-46s bench.py failed (a stale variant built before an API change? rebuild it)' % '${lib:-in-tree}')"
done
