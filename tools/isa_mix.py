"""
isa_mix.py -- instruction mix of one blind-rotate iteration of k_bootstrap<1>, counted in the ISA.

    python tools/isa_mix.py            # compiles nufhe_amd/csrc/kernels.hip to assembly (hipcc -S)

Prints VGPR / scratch use and, for the body of the per-iteration loop, the number of VALU, SALU, LDS,
memory, s_waitcnt and s_nop instructions (inner loops weighted by their trip counts where the
compiler kept them rolled).  The VALU figure is the one that decides the kernel's speed.
`tools/isa_pieces.hip` compiles the building blocks (16-point pass, twiddle layers, paired MAC ...)
as separate kernels for the same kind of count:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Inufhe_amd/csrc -S --cuda-device-only tools/isa_pieces.hip -o /tmp/p.s
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'nufhe_amd', 'csrc')


def main():
    asm = '/tmp/nufhe_kernels.s'
    out = subprocess.run(
        ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', 'kernels.hip', '-o', asm,
         '-Rpass-analysis=kernel-resource-usage'] + sys.argv[1:], cwd=CSRC, capture_output=True, text=True)
    text = out.stderr
    m = re.search(r'Function Name: _Z11k_bootstrapILi1EEv8BrLaunch.*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)', text, re.S)
    if m:
        print('k_bootstrap<1>: VGPRs %s, scratch %s B' % m.groups())
    lines = open(asm).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z11k_bootstrapILi1EEv8BrLaunch:'))
    end = next(i for i, l in enumerate(lines) if i > start and 's_endpgm' in l)
    lines = lines[start:end]
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    # the iteration loop = the largest backward-branch span or the Depth=1 header after the prologue
    hdr = [i for i, l in enumerate(lines) if 'Loop Header: Depth=1' in l and i > 300]
    body_start = hdr[-1] if hdr else 0
    inner = sorted([lp for lp in loops if lp[0] > body_start and lp[1] - lp[0] > 1000], key=lambda x: x[1] - x[0])
    body_end = next((i for i, l in enumerate(lines) if i > body_start + 3000 and re.match(r'^\.LBB\d+_\d+:\s*$', l)
                     and not any(a <= i <= b for a, b in inner)), len(lines) - 1)

    def weight(i):
        w = 1
        for k, (a, b) in enumerate(inner):
            if a <= i <= b:
                w *= 2          # rolled m- or d-loop: two trips each
        return w

    c = collections.Counter()
    for i in range(body_start, body_end):
        mm = re.match(r'^\s+([a-z_0-9]+)', lines[i])
        if not mm:
            continue
        op, w = mm.group(1), weight(i)
        if op == 's_nop':
            c['s_nop'] += w
            c['nop_cycles'] += w * (int(lines[i].split()[1]) + 1)
        elif op.startswith('v_'):
            c['valu'] += w
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += w
        elif op.startswith('s_'):
            c['salu'] += w
        elif op.startswith('ds_'):
            c['lds'] += w
        elif op.startswith(('global', 'scratch', 'buffer')):
            c['vmem'] += w
    print('per blind-rotate iteration (rolled inner loops: %d):' % len(inner), dict(c))


if __name__ == '__main__':
    main()
