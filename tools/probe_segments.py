"""probe_segments.py -- with a BR_PROBE variant library (tools/build_variant.sh probe -DBR_PROBE; run with
NUFHE_HIP_LIBRARY=gpurun_variants/libnufhe_hip_probe.so): shader-clock ticks a wave spends (average over all waves) in the
segments of the external product, per blind-rotate iteration, for a 4096-bit NAND."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import nufhe_amd as nufhe
from nufhe_amd import _lib

TRANSFORM = sys.argv[1] if len(sys.argv) > 1 else 'NTT'
ctx = nufhe.Context(rng=nufhe.DeterministicRNG(123))
secret, cloud = ctx.make_key_pair(transform_type=TRANSFORM)
vm = ctx.make_virtual_machine(cloud)
rs = numpy.random.RandomState(3)
bits = 4096
a = ctx.encrypt(secret, rs.randint(0, 2, bits).astype(bool))
b = ctx.encrypt(secret, rs.randint(0, 2, bits).astype(bool))
lib = _lib.lib()
lib.nufhe_probe_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)()
vm.gate_nand(a, b)
lib.nufhe_probe_read(buf)
reps = 3
for _ in range(reps):
    vm.gate_nand(a, b)
lib.nufhe_probe_read(buf)
if TRANSFORM == 'FFT':
    names_fft = ['fwd pair m=0', 'mac m=0', 'fwd pair m=1', 'mac m=1', 'inverse pair', 'rotate + park', 'round + update acc']
names = ['digits+fwd x2 (m=0)', 'mac m=0 (canonical)', 'digits+fwd x2 (m=1)', 'mac_l4 mo=0', 'inverse mo=0 + acc',
         'mac_l4 mo=1', 'inverse mo=1 + acc', 'wave sync + loop']
if TRANSFORM == 'FFT':
    names = names_fft
# block 0 hosts bits 0..7 in round one; its wave 0 runs ~500 iterations per gate (a == 0 skipped); blocks of the second
# round have other indices, so the count is per gate
tot = 0
waves = buf[13] / reps
iters = 500.0 * waves
print('blind-rotate wall time per wave: %.3f ms -> shader clock %.3f GHz' % (buf[12] / max(1, buf[13]) * 1e-5, buf[14] / max(1, buf[12]) * 0.1))
print('waves per gate:', waves, ' blind-rotate ticks per wave and iteration: %.0f' % (buf[14] / max(1, buf[13]) / 500.0))
for n, v in zip(names, buf):
    per = v / reps / iters
    tot += per
    print('%-24s %9.0f ticks per iteration' % (n, per))
print('%-24s %9.0f' % ('sum', tot))

life = (ctypes.c_uint * (2 * 4096))()
lib.nufhe_probe_lifetimes.argtypes = [ctypes.POINTER(ctypes.c_uint), ctypes.c_int]
lib.nufhe_probe_lifetimes(life, 2 * 4096)
t = numpy.array(life, dtype=numpy.int64).reshape(4096, 2)
t0 = t[:, 0].min()
start = (t[:, 0] - t0) * 1e-5
end = (t[:, 1] - t0) * 1e-5
dur = end - start
wave = numpy.arange(4096) % 8
block = numpy.arange(4096) // 8
print('last gate: wave start / end (ms since the first start), by wave index in the work-group')
for w in range(8):
    m = wave == w
    print('  wave %d: lifetime mean %.2f min %.2f max %.2f ms; start mean %.2f, end mean %.2f max %.2f' % (
        w, dur[m].mean(), dur[m].min(), dur[m].max(), start[m].mean(), end[m].mean(), end[m].max()))
first = start < 1.0
print('first-round work-groups: %d waves, end mean %.2f max %.2f; second round: start mean %.2f min %.2f, end max %.2f' % (
    first.sum(), end[first].mean(), end[first].max(), start[~first].mean(), start[~first].min(), end[~first].max()))
bd = numpy.array([dur[block == b].max() - dur[block == b].min() for b in range(512)])
print('spread of lifetimes inside a work-group: mean %.2f max %.2f ms' % (bd.mean(), bd.max()))
