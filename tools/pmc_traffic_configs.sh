#!/bin/bash
# pmc_traffic_configs.sh TAG -- HBM traffic of the bootstrap kernel for the BASELINE configurations that tools/profile.sh
# does not cover (it profiles NAND at 4096 bits): MUX at 4096 bits and NAND at 2048 bits, both transforms.  Two counter
# passes per configuration (FETCH_SIZE and WRITE_SIZE each in its own run: MI355X_MICROARCH.md), `bench.py --steps 2
# --warmup 1`.  Merges the result into gpurun_out/prof_TAG/pmc_traffic.json under "configs" (keys TRANSFORM/gate/bits);
# copy that file to profiles/pmc_traffic.json -- bench.py's `roofline.traffic` reads it.
TAG=${1:-run}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for CFG in ${CONFIGS:-NTT/mux/4096 FFT/mux/4096 NTT/nand/2048 FFT/nand/2048}; do
    IFS=/ read TR GATE BITS <<< "$CFG"
    for C in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/traffic_${TR}_${GATE}_${BITS}_$C" -- \
            python "$ROOT/bench.py" --steps 2 --warmup 1 --transform $TR --gate $GATE --bits $BITS --no-extra --no-cpu-baseline \
            > /dev/null 2> "$OUT/traffic_${TR}_${GATE}_${BITS}_$C.log"
    done
done
python - "$OUT" "$ROOT" <<'PY'
import glob, json, os, sys
out, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(root, 'tools'))
import profile_summary as ps
path = os.path.join(out, 'pmc_traffic.json')
base = json.load(open(path)) if os.path.exists(path) else json.load(open(os.path.join(root, 'profiles', 'pmc_traffic.json')))
configs = base.setdefault('configs', {})
for d in sorted(glob.glob(os.path.join(out, 'traffic_*_FETCH_SIZE'))):
    tr, gate, bits = os.path.basename(d).split('_')[1:4]
    kern = ps.KERNELS[tr]
    f, _ = ps.pass_counters(d, kern)
    w, _ = ps.pass_counters(d.replace('FETCH_SIZE', 'WRITE_SIZE'), kern)
    if 'FETCH_SIZE' not in f or 'WRITE_SIZE' not in w:
        print('missing counters for', d)
        continue
    configs['%s/%s/%s' % (tr, gate, bits)] = {
        'kernel': kern, 'workload': 'gate_%s %s bits' % (gate, bits),
        'fetch_size_kb': f['FETCH_SIZE'], 'write_size_kb': w['WRITE_SIZE'],
        'hbm_bytes_per_launch': (2 * f['FETCH_SIZE'] + w['WRITE_SIZE']) * 1024,
        'note': 'tools/pmc_traffic_configs.sh: FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes of `bench.py '
                '--steps 2 --warmup 1`; FETCH_SIZE doubled per MI355X_MICROARCH.md'}
json.dump(base, open(path, 'w'), indent=1, sort_keys=True)
print(json.dumps(configs, indent=1))
PY
