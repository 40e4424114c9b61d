"""valu_class_costs.py MICROBENCH.txt -> profiles/valu_class_costs.json

Average issue cost (shader cycles per wave64 instruction per SIMD, 2 waves per SIMD, from clock64) of the
two integer VALU issue classes tools/isa_mix.py counts, taken from the output of tools/microbench_l4
(profiles/r02_microbench_l4.txt).  bench.py multiplies the ISA counts of the bootstrap kernel by these."""
import json
import os
import re
import sys

FULL = ['v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_not_b32', 'v_mov_b32',
        'v_ashrrev_i32 (const)', 'v_lshrrev_b32 (vgpr amount)']
HALF = ['v_perm_b32', 'v_lshl_add_u32', 'v_add3_u32', 'v_and_or_b32', 'v_mad_i32_i24', 'v_bfm_b32', 'v_bfe_i32',
        'v_lshlrev_b32 (const)', 'v_mad_u64_u32', 'v_lshl_add_u64', 'v_addc_co_u32']


def main(path):
    vals = {}
    for line in open(path):
        m = re.match(r'^(v_\S+(?: \([^)]*\))?)\s+[\d.]+ ms\s+([\d.]+) clock64 ticks', line)
        if m:
            vals[m.group(1)] = float(m.group(2))
    full = [vals[k] for k in FULL if k in vals]
    half = [vals[k] for k in HALF if k in vals]
    out = {'full_rate_cycles': sum(full) / len(full), 'half_rate_cycles': sum(half) / len(half),
           'f64_fma_cycles': vals.get('v_fma_f64'),
           'full_rate_ops': {k: vals[k] for k in FULL if k in vals}, 'half_rate_ops': {k: vals[k] for k in HALF if k in vals},
           'other_ops': {k: v for k, v in vals.items() if k not in FULL and k not in HALF},
           'source': os.path.basename(path),
           'note': 'shader cycles (clock64) per wave64 instruction per SIMD with 2 waves per SIMD, 16 independent '
                   'chains per lane (tools/microbench_l4.hip); class = plain average of the listed opcodes'}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    json.dump(out, open(os.path.join(root, 'profiles', 'valu_class_costs.json'), 'w'), indent=1, sort_keys=True)
    print(json.dumps({k: out[k] for k in ('full_rate_cycles', 'half_rate_cycles', 'f64_fma_cycles')}))


if __name__ == '__main__':
    main(sys.argv[1])
