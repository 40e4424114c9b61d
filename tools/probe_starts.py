import ctypes, os, sys, numpy
sys.path.insert(0, '/root/repo')
import nufhe_amd as nufhe
from nufhe_amd import _lib
TR = sys.argv[1]
ctx = nufhe.Context(rng=nufhe.DeterministicRNG(123))
secret, cloud = ctx.make_key_pair(transform_type=TR)
vm = ctx.make_virtual_machine(cloud)
rs = numpy.random.RandomState(3)
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
a = ctx.encrypt(secret, rs.randint(0, 2, bits).astype(bool)); b = ctx.encrypt(secret, rs.randint(0, 2, bits).astype(bool))
lib = _lib.lib()
for _ in range(3): vm.gate_nand(a, b)
life = (ctypes.c_uint * (2 * 8192))()
lib.nufhe_probe_lifetimes.argtypes = [ctypes.POINTER(ctypes.c_uint), ctypes.c_int]
lib.nufhe_probe_lifetimes(life, 2 * 8192)
t = numpy.array(life, dtype=numpy.int64).reshape(8192, 2)[:bits]
t0 = t[:, 0].min()
start = (t[:, 0] - t0) * 1e-5; end = (t[:, 1] - t0) * 1e-5
print(TR, bits, 'bits: wave starts (ms after the first): percentiles 0/10/50/90/100 = %s' % numpy.round(numpy.percentile(start, [0, 10, 50, 90, 100]), 3))
print('   ends: %s ; life: %s' % (numpy.round(numpy.percentile(end, [0, 10, 50, 90, 100]), 3), numpy.round(numpy.percentile(end - start, [0, 50, 100]), 3)))
buf = (ctypes.c_ulonglong * 16)()
lib.nufhe_probe_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
lib.nufhe_profile_enable(ctx.thread.handle, 1)
lib.nufhe_probe_read(buf)
vm.gate_nand(a, b)
br = ctypes.c_float(); ks = ctypes.c_float()
lib.nufhe_profile_last(ctx.thread.handle, ctypes.byref(br), ctypes.byref(ks))
lib.nufhe_probe_read(buf)
e, b0, e0, x = buf[8], buf[9], buf[10], buf[11]
print('   WG 0 wave 0: entry -> blind-rotate begin %.3f ms, blind rotate %.3f ms, end -> exit %.3f ms; kernel (HIP events) %.3f ms' % (
    (b0 - e) * 1e-5, (e0 - b0) * 1e-5, (x - e0) * 1e-5, br.value))
