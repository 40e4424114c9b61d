"""check_soffset_war.py -- scans the gfx950 ISA of every kernel translation unit for the pattern that gave wrong words in
round 6 (profiles/r06_xfft_experiments.txt, 4b): a buffer instruction that reads an SGPR as its scalar offset, with a scalar
instruction REWRITING that SGPR within the next few instructions.  With two waves per SIMD the rewritten value reached the
memory instruction.  Exit status 1 when a candidate is found.

    python tools/check_soffset_war.py [window=4]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'nufhe_amd', 'csrc')
UNITS = {'kernels.hip': [], 'kernels_xfft.hip': [], 'kernels_team.hip': ['-DFF_MULWIDE_PLAIN'],
         'kernels_team8.hip': ['-DFF_MULWIDE_PLAIN', '-mllvm', '-amdgpu-sched-strategy=max-ilp']}


def scan(lines, window):
    found = []
    func = None
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            func = m.group(1)
        m = re.match(r'\s+buffer_(?:load|store|atomic)\w* .*\], (s\d+)(?: |$)', l)
        if not m:
            continue
        reg = m.group(1)
        seen = 0
        for j in range(i + 1, len(lines)):
            t = lines[j].split(';')[0].strip()
            if not t or t.endswith(':') or t.startswith('.'):
                continue
            seen += 1
            if seen > window:
                break
            w = re.match(r's_\w+ (s\d+|s\[(\d+):(\d+)\])', t)
            if w and not t.startswith(('s_waitcnt', 's_nop', 's_cmp', 's_cbranch', 's_branch', 's_barrier', 's_setprio')):
                n = int(reg[1:])
                if w.group(1) == reg or (w.group(2) and int(w.group(2)) <= n <= int(w.group(3))):
                    found.append((func, i + 1, l.strip()[:80], t))
    return found


def main():
    window = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    total = 0
    for unit, flags in UNITS.items():
        asm = '/tmp/war_%s.s' % unit.replace('.hip', '')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only',
                               unit, '-o', asm] + flags, cwd=CSRC, stderr=subprocess.DEVNULL)
        lines = open(asm).read().split('\n')
        nbuf = sum(1 for l in lines if re.match(r'\s+buffer_', l))
        found = scan(lines, window)
        print('%-20s %5d buffer instructions, %d candidates' % (unit, nbuf, len(found)))
        for f in found:
            print('   %s line %d: %s  <-  %s' % f)
        total += len(found)
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main())
