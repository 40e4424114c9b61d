"""xfft_loop_events.py [asm] -- where the scratch / global / buffer-store instructions of k_bootstrap_xfft's iteration loop
sit (line offsets inside the loop body), for register-pressure work on blind_rotate_xfft.h."""
import re
import subprocess
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = '/tmp/xfft.s'
flags = sys.argv[1:]
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only',
                       'kernels_xfft.hip', '-o', asm] + flags, cwd=os.path.join(ROOT, 'nufhe_amd', 'csrc'),
                      stderr=subprocess.DEVNULL)
L = open(asm).read().split('\n')
s = next(i for i, l in enumerate(L) if l.startswith('_Z16k_bootstrap_xfft8BrLaunch:'))
e = next(i for i, l in enumerate(L) if i > s and 's_endpgm' in l)
body = L[s:e]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
best = None
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        sp = (labels[m.group(1)], i)
        if best is None or sp[1] - sp[0] > best[1] - best[0]:
            best = sp
loop = body[best[0]:best[1] + 1]
print('loop lines', len(loop))
bl = [i for i, l in enumerate(loop) if l.strip().startswith('buffer_load')]
print('buffer loads: first', bl[:2], 'last', bl[-2:], 'count', len(bl))
for i, l in enumerate(loop):
    t = l.strip()
    if t.startswith(('scratch_', 'global_', 'buffer_store')) or 's_barrier' in t:
        print(i, t[:100])
# a scalar offset register rewritten within 3 instructions of the VMEM instruction that reads it (gave wrong words with two
# waves per SIMD in round 6: blind_rotate_xfft.h, brx_park_store)
bad = 0
for i, l in enumerate(loop):
    m = re.match(r'\s+buffer_(load|store)\w+ .*\], (s\d+) offen', l)
    if m:
        for j in range(i + 1, min(i + 4, len(loop))):
            if re.match(r'\s+s_mov\w+ %s,' % m.group(2), loop[j]):
                bad += 1
                print('soffset rewritten behind its reader:', i, l.strip()[:70], '|', loop[j].strip())
print('soffset WAR candidates:', bad)
