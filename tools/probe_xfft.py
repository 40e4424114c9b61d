"""probe_xfft.py -- with a BR_PROBE variant of the exact-FFT unit (tools/build_xfft_variant.sh probe -DBR_PROBE; run with
NUFHE_HIP_LIBRARY=gpurun_variants/libnufhe_hip_probe.so): shader-clock ticks a wave spends in the segments of
brx_step, per blind-rotate iteration, average over all waves of a 4096-bit NAND."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import nufhe_amd as nufhe
from nufhe_amd import _lib

ctx = nufhe.Context(rng=nufhe.DeterministicRNG(123))
secret, cloud = ctx.make_key_pair()
cloud.set_engine('exact-fft')
vm = ctx.make_virtual_machine(cloud)
rs = numpy.random.RandomState(3)
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = ctx.encrypt(secret, rs.randint(0, 2, bits).astype(bool))
b = ctx.encrypt(secret, rs.randint(0, 2, bits).astype(bool))
lib = _lib.lib()
lib.nufhe_probe_read_xfft.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)()
vm.gate_nand(a, b)
lib.nufhe_probe_read_xfft(buf)
reps = 3
for _ in range(reps):
    vm.gate_nand(a, b)
lib.nufhe_probe_read_xfft(buf)
names = ['rotate + park store', 'digits + fwd pair m=0', 'mac m=0 (64 key loads)', 'digits + fwd pair m=1', 'mac m=1 (64 key loads)',
         'inverse pair lo + park load + round', 'inverse pair hi', 'round hi + mirror write']
if bits <= 512:     # the four-waves-per-bit kernel (brxq_step): every wave of a team is counted
    names = ['key requests + rotation + digits', 'forward transform', 'X write + team barrier 1', 'X reads + products',
             'team barrier 2', 'inverse transform', 'rounding + ACC atomics', 'team barrier 3']
waves = buf[13] / reps
iters = 500.0 * waves
print('blind-rotate wall time per wave: %.3f ms -> shader clock %.3f GHz' % (buf[12] / max(1, buf[13]) * 1e-5, buf[14] / max(1, buf[12]) * 0.1))
print('waves per gate:', waves, ' blind-rotate ticks per wave and iteration: %.0f' % (buf[14] / max(1, buf[13]) / 500.0))
tot = 0
for n, v in zip(names, buf):
    per = v / reps / iters
    tot += per
    print('%-40s %9.0f ticks per iteration' % (n, per))
print('%-40s %9.0f' % ('sum', tot))
