"""probe_team8.py -- with a BR_PROBE variant library: shader ticks wave 0 of work-group 0 spends in the phases of one step
of the 8-wave half-ring team kernel (k_bootstrap_team8), per blind-rotate iteration."""
import ctypes, os, sys, numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nufhe_amd as nufhe
from nufhe_amd import _lib
ctx = nufhe.Context(rng=nufhe.DeterministicRNG(123))
secret, cloud = ctx.make_key_pair()
vm = ctx.make_virtual_machine(cloud)
rs = numpy.random.RandomState(3)
bits = 64
a = ctx.encrypt(secret, rs.randint(0, 2, bits).astype(bool)); b = ctx.encrypt(secret, rs.randint(0, 2, bits).astype(bool))
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 8)()
lib.nufhe_probe_read_team8.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
vm.gate_nand(a, b); ctx.thread.synchronize()
lib.nufhe_probe_read_team8(buf)
reps = 4
for _ in range(reps): vm.gate_nand(a, b)
ctx.thread.synchronize()
lib.nufhe_probe_read_team8(buf)
names = ['key loads issued + rotate + digits', 'half forward transform', 'products -> LDS', 'barrier 1', 'reduce partial sums',
         'half inverse transform + join store', 'barrier 2', 'join + ACC update + barrier 3']
tot = 0
for n, v in zip(names, buf):
    per = v / reps / 500.0
    tot += per
    print('%-40s %8.0f ticks per iteration' % (n, per))
print('%-40s %8.0f (= %.2f us at 2.4 GHz)' % ('sum', tot, tot / 2400.0))
