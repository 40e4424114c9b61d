#!/bin/bash
# usage: run on GPU box: prints kernel_ms, wave_ms, clock for each variant
cd /root/repo
for lib in "" gpurun_variants/libnufhe_hip_*.so; do
  NUFHE_HIP_LIBRARY=${lib:+/root/repo/$lib} python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-44s step %.3f K1 %.3f wave_ms %.3f clock %.3f correct %s' % ('${lib:-in-tree}', d['ms_per_step'], r['kernel_ms'], r.get('wave_ms_in_kernel',0), r.get('clock_ghz_in_kernel',0), d['correct']))"
done
