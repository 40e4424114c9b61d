"""valu_issue_costs.py MICROBENCH.txt -> profiles/valu_issue_costs.json

Issue cost of the VALU classes on gfx950 from tools/microbench_issue (profiles/r03_microbench_issue.txt): shader
cycles per wave64 instruction per SIMD = launch duration x in-kernel clock / instructions per SIMD, at 1, 2, 3, 4 and 8
waves per SIMD.  Two classes matter for the bootstrap kernels:

  plain   VOP1/VOP2 v_add/sub/subrev_u32, v_and/or/xor/not_b32, v_mov_b32, v_lshrrev_b32, v_ashrrev_i32 with VGPR,
          inline-constant or literal operands: ~2.2 cycles from 2 waves per SIMD on -- two of them from DIFFERENT waves
          share one issue slot; a lone wave pays 4.4
  other   everything else (VOP3, SDWA, DPP, v_lshlrev_b32, multiplies, v_mad_u64_u32, carries, an SGPR operand, fp32/
          fp64): 4.1-4.4 cycles at ANY occupancy -- one instruction per ~4.2-cycle issue slot

A stream that mixes the classes pays ~4 cycles for EVERY instruction at 2 waves per SIMD ("blend" lines): plain
instructions only pair when both waves have one at their head.  bench.py reads this file."""
import json
import os
import re
import sys

PLAIN = ['v_add_u32', 'v_sub_u32', 'v_xor_b32', 'v_mov_b32', 'v_lshrrev_b32 const', 'v_ashrrev_i32 const']
OTHER = ['v_add_u32 sgpr', 'v_lshlrev_b32 const', 'v_lshl_add_u32', 'v_add3_u32', 'v_sad_u32', 'v_perm_b32', 'v_bfe_i32',
         'v_alignbit_b32', 'v_and_or_b32', 'v_lshl_or_b32', 'v_add_u32_sdwa', 'v_mov_b32_dpp', 'v_pk_add_u16',
         'v_mad_u64_u32', 'v_lshl_add_u64', 'v_addc_co_u32', 'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mul_u32_u24',
         'v_mad_u32_u24', 'v_mad_i32_i24', 'v_dot4_i32_i8']
COLS = ['1', '2', '3', '4', '8']


def main(path):
    rows = {}
    for line in open(path):
        m = re.match(r'^(.+?)\s{2,}([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+\| dependent chain:\s+([\d.]+)\s+([\d.]+)', line)
        if m:
            rows[m.group(1).strip()] = dict(zip(COLS, (float(m.group(i)) for i in range(2, 7))),
                                            dependent_chain_1_wave=float(m.group(7)))
    def avg(names, col):
        v = [rows[n][col] for n in names if n in rows]
        return sum(v) / len(v)
    out = {
        'source': os.path.basename(path),
        'unit': 'shader cycles per wave64 instruction per SIMD (launch duration x in-kernel clock / instructions per SIMD)',
        'plain_cycles': {c: avg(PLAIN, c) for c in COLS},
        'other_cycles': {c: avg(OTHER, c) for c in COLS},
        'literal_cycles': dict(rows.get('v_add_u32 literal', {})),
        'fma_f32_cycles': dict(rows.get('v_fma_f32', {})),
        'fma_f64_cycles': dict(rows.get('v_fma_f64', {})),
        'mad_u64_u32_cycles': dict(rows.get('v_mad_u64_u32', {})),
        'blend_3_plain_2_other_cycles': dict(rows.get('blend 3 add : 2 lshl_add', {})),
        'dependent_chain_one_wave_cycles': rows.get('v_add_u32', {}).get('dependent_chain_1_wave'),
        'barrier_aligned_runs': {k: v for k, v in rows.items() if k.startswith('barrier,')},
        'alternating_runs': {k: v for k, v in rows.items() if re.match(r'^\d+ add : ', k)},
        'plain_ops': PLAIN, 'other_ops': OTHER,
        # the rates bench.py prices against: the best each class reaches on the machine (8 waves per SIMD) ...
        'machine_plain_cycles': avg(PLAIN, '8'), 'machine_other_cycles': avg(OTHER, '8'),
        # ... and what the classes cost at the bootstrap kernels' occupancy (2 waves per SIMD, 256 VGPRs)
        'two_wave_plain_cycles': avg(PLAIN, '2'), 'two_wave_other_cycles': avg(OTHER, '2'),
        'nominal_cycles_microarch_guide': 2.0,
        'note': 'MI355X_MICROARCH.md quotes 2 cycles per wave64 VALU instruction (SIMD-32); measured here, that rate '
                'is reached by the plain class only and only when two waves pair; v_fma_f32 itself issues at 3.7-4.0',
    }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    json.dump(out, open(os.path.join(root, 'profiles', 'valu_issue_costs.json'), 'w'), indent=1, sort_keys=True)
    print(json.dumps({k: out[k] for k in ('machine_plain_cycles', 'machine_other_cycles', 'two_wave_plain_cycles',
                                          'two_wave_other_cycles')}))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r03_microbench_issue.txt'))
