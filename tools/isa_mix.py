"""
isa_mix.py -- instruction mix of one blind-rotate iteration of the bootstrap kernels, counted in the ISA.

    python tools/isa_mix.py            # compiles nufhe_amd/csrc/kernels.hip / kernels_xfft.hip to assembly (hipcc -S) and prints
    python tools/isa_mix.py --write    # ... and refreshes profiles/isa_mix.json (read by bench.py)

For the body of the per-iteration loop of k_bootstrap<1> (NTT), k_bootstrap_fft and k_bootstrap_xfft it reports VGPR /
scratch use and the number of VALU instructions by ISSUE CLASS (measured on gfx950 at the kernel's
occupancy, profiles/r02_microbench_l4.txt):
    full   v_add/sub/subrev_u32, v_and/or/xor/not_b32, v_lshrrev_b32, v_ashrrev_i32, v_mov_b32, v_cndmask_b32
    half   everything else 32/64-bit integer (carry chains, compares, v_lshlrev, v_perm, v_mad_u64_u32, VOP3 ...)
    f64    v_fma_f64 / v_fmac_f64 / v_mul_f64 / v_add_f64 (counted separately: flops)
plus SALU, LDS, memory, s_waitcnt and hazard s_nop counts.  The weighted VALU figure decides the NTT
kernel's speed.  `profiles/isa_mix.json` also records a hash of the device sources so that
tests/test_cabi_and_host.py can tell when it is stale.
"""
import collections
import hashlib
import importlib.util
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'nufhe_amd', 'csrc')
FULL_RATE = {'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_not_b32',
             'v_lshrrev_b32', 'v_ashrrev_i32', 'v_mov_b32', 'v_cndmask_b32'}
F64 = {'v_fma_f64': 2, 'v_fmac_f64': 2, 'v_mul_f64': 1, 'v_add_f64': 1}
KERNELS = {'k_bootstrap<1>': ('kernels.hip', '_Z11k_bootstrapILi1EEv8BrLaunch'),
           'k_bootstrap_fft': ('kernels.hip', '_Z15k_bootstrap_fft8BrLaunch'),
           'k_bootstrap_xfft': ('kernels_xfft.hip', '_Z16k_bootstrap_xfft8BrLaunch')}


def source_hash():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith(('.h', '.hip')) and name != 'api.hip':     # api.hip is the host-side translation unit: no device code
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), 'rb').read())
    return h.hexdigest()[:16]


def _asm_sched():
    spec = importlib.util.spec_from_file_location('asm_sched', os.path.join(ROOT, 'tools', 'asm_sched.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def strip(op):
    return re.sub(r'_(e32|e64|sdwa|dpp)$', '', op)


def loop_body(lines):
    """lines of the per-iteration loop = the longest span closed by a backward branch"""
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            span = (labels[mm.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    return lines[best[0]:best[1] + 1]


def count(body):
    c = collections.Counter()
    sched = _asm_sched()
    for l in body:
        t = l.split(';')[0].strip()
        if t.startswith('v_'):
            # issue classes of profiles/valu_issue_costs.json: "plain" = the VOP1/VOP2 instructions that can share an
            # issue slot with another wave's plain instruction (no SGPR / VCC operand, no SDWA / DPP / VOP3 form)
            c['valu_plain' if sched.Inst(t, 0).plain else 'valu_other'] += 1
    for l in body:
        mm = re.match(r'^\s+([a-z_0-9]+)', l)
        if not mm:
            continue
        raw = mm.group(1)
        op = strip(raw)
        if raw.endswith(('_sdwa', '_dpp')):
            op += '_sdwa'          # sub-dword / cross-lane operand forms issue at the VOP3 rate (microbench: 4.4 cycles)
        if op == 's_nop':
            c['s_nop'] += 1
            c['nop_wait_states'] += int(l.split()[1]) + 1
        elif op in F64:
            c['valu'] += 1
            c['valu_f64'] += 1
            c['f64_flops_per_lane'] += F64[op]
        elif op.startswith('v_'):
            c['valu'] += 1
            c['valu_full_rate' if op in FULL_RATE else 'valu_half_rate'] += 1
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith('scratch'):
            c['scratch'] += 1
        elif op.startswith(('global', 'buffer')):
            c['vmem'] += 1
    return dict(c)


def main():
    args = [a for a in sys.argv[1:] if a != '--write']
    result = {'source_hash': source_hash(),
              'note': 'python tools/isa_mix.py --write; per blind-rotate iteration of one wave (one bit)'}
    compiled = {}
    for name, (src, sym) in KERNELS.items():
        if src not in compiled:
            asm = '/tmp/nufhe_%s.s' % src.replace('.hip', '')
            out = subprocess.run(
                ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', src, '-o', asm,
                 '-Rpass-analysis=kernel-resource-usage'] + args, cwd=CSRC, capture_output=True, text=True)
            if out.returncode:
                sys.exit(out.stderr[-3000:])
            compiled[src] = (out.stderr, open(asm).read().split('\n'))
        text, all_lines = compiled[src]
        m = re.search(r'Function Name: ' + re.escape(sym) + r'\b.*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)', text, re.S)
        start = next(i for i, l in enumerate(all_lines) if l.startswith(sym + ':'))
        end = next(i for i, l in enumerate(all_lines) if i > start and 's_endpgm' in l)
        c = count(loop_body(all_lines[start:end]))
        c['vgprs'] = int(m.group(1)) if m else None
        c['scratch_bytes'] = int(m.group(2)) if m else None
        result[name] = c
        print(name, json.dumps(c))
    if '--write' in sys.argv:
        json.dump(result, open(os.path.join(ROOT, 'profiles', 'isa_mix.json'), 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
