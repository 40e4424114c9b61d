"""
isa_lines.py -- where the instructions of the blind-rotate loop come from, by source function.

    python tools/isa_lines.py [kernel-symbol [source.hip]] [-v] [-DMACRO...]     (default: k_bootstrap<1> of kernels.hip)

Compiles kernels.hip to assembly with line tables (-gline-tables-only), takes the per-iteration loop of the
kernel (as tools/isa_mix.py does) and attributes every instruction to the innermost source function its
`.loc` names.  Prints, per function, the VALU count by issue class, the issue cycles they cost
(profiles/valu_class_costs.json) and the LDS / hazard-nop counts.  A static profile: it says where issue
slots go, not where the waves wait.
"""
import collections
import json
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_mix  # noqa: E402

ROOT = isa_mix.ROOT
CSRC = isa_mix.CSRC


def function_map(path):
    """line number -> name of the function defined around it (top-level definitions only)"""
    names = {}
    cur = None
    pending = False
    for no, line in enumerate(open(path, errors='replace').read().split('\n'), 1):
        if re.match(r'^(FF_FN|FF_HD|template|static|__device__|__global__|inline)', line):
            pending = True
        if pending:
            mm = re.search(r'\b([A-Za-z_]\w*)\s*\(', line)
            if mm and mm.group(1) not in ('__launch_bounds__', '__attribute__', 'defined', 'aligned'):
                cur = mm.group(1)
                pending = False
        names[no] = cur
    return names


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith('-')]
    want = pos[0] if pos else 'k_bootstrap<1>'
    unit, sym = isa_mix.KERNELS.get(want, ('kernels.hip', want))
    src = os.path.abspath(pos[1]) if len(pos) > 1 else os.path.join(CSRC, unit)
    asm = '/tmp/nufhe_kernels_g.s'
    out = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-I' + CSRC,
                          '-gline-tables-only', src, '-o', asm] + [a for a in sys.argv[1:] if a.startswith('-D')],
                         cwd=CSRC, capture_output=True, text=True)
    if out.returncode:
        sys.exit(out.stderr[-3000:])
    lines = open(asm).read().split('\n')
    files = {}
    for l in lines:
        mm = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if mm:
            d, f = (mm.group(2), mm.group(3)) if mm.group(3) is not None else ('.', mm.group(2))
            files[int(mm.group(1))] = os.path.normpath(os.path.join(CSRC if d == '.' else d, f))
    fmaps = {}
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ':'))
    end = next(i for i, l in enumerate(lines) if i > start and 's_endpgm' in l)
    body = isa_mix.loop_body(lines[start:end])
    costs = json.load(open(os.path.join(ROOT, 'profiles', 'valu_class_costs.json')))
    c_full = costs.get('full_rate_cycles', 3.0)
    c_half = costs.get('half_rate_cycles', 4.4)
    per = collections.defaultdict(collections.Counter)
    where = ('?', 0)
    for l in body:
        mm = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
        if mm:
            where = (files.get(int(mm.group(1)), '?'), int(mm.group(2)))
            continue
        mm = re.match(r'^\s+([a-z_0-9]+)', l)
        if not mm:
            continue
        op = isa_mix.strip(mm.group(1))
        if mm.group(1).endswith(('_sdwa', '_dpp')):
            op += '_sdwa'
        path, no = where
        if path not in fmaps:
            fmaps[path] = function_map(path) if os.path.exists(path) else {}
        key = '%s:%s' % (os.path.basename(path), fmaps[path].get(no) or '?')
        c = per[key]
        if op == 's_nop':
            c['nop_states'] += int(l.split()[1]) + 1
        elif op.startswith('v_'):
            cls = 'full' if op in isa_mix.FULL_RATE else 'half'
            c[cls] += 1
            c['op:' + op] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif op.startswith(('global', 'buffer', 'scratch')):
            c['vmem'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
    rows = []
    for key, c in per.items():
        cyc = c['full'] * c_full + c['half'] * c_half + c['nop_states']
        rows.append((cyc, key, c))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print('%-44s %7s %7s %7s %6s %6s %8s %6s' % ('function', 'full', 'half', 'nop', 'lds', 'wait', 'cycles', '%'))
    for cyc, key, c in rows:
        print('%-44s %7d %7d %7d %6d %6d %8.0f %6.1f' % (key, c['full'], c['half'], c['nop_states'], c['lds'],
                                                      c['waitcnt'], cyc, 100 * cyc / tot))
        if '-v' in sys.argv:
            ops = sorted(((n, o[3:]) for o, n in c.items() if o.startswith('op:')), reverse=True)[:8]
            print('      ' + ', '.join('%s %d' % (o, n) for n, o in ops))
    print('%-44s %7d %7d %7d %6d %6d %8.0f' % ('total', sum(r[2]['full'] for r in rows), sum(r[2]['half'] for r in rows),
                                          sum(r[2]['nop_states'] for r in rows), sum(r[2]['lds'] for r in rows),
                                          sum(r[2]['waitcnt'] for r in rows), tot))


if __name__ == '__main__':
    main()
