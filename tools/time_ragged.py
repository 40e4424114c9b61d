"""NAND timing at batch sizes around whole rounds of the one-wave kernels (8 x CUs bits):
`python tools/time_ragged.py [native | exact-fft | FFT]` (FFT: an FFT key)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch, nufhe_amd
engine = sys.argv[1] if len(sys.argv) > 1 else 'exact-fft'
ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(1))
sk, ck = ctx.make_key_pair(transform_type='FFT' if engine == 'FFT' else 'NTT')
if engine != 'FFT':
    ck.set_engine(engine)
vm = ctx.make_virtual_machine(ck)
out = {}
for B in (2048, 2060, 2304, 2560, 3000, 3072, 4096, 4100, 4352, 4608, 5000, 6144, 6200):
    m = numpy.random.RandomState(B).randint(0, 2, size=B).astype(bool)
    a = ctx.encrypt(sk, m); b = ctx.encrypt(sk, ~m)
    r = vm.gate_nand(a, b); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        r = vm.gate_nand(a, b)
    torch.cuda.synchronize()
    out[B] = round((time.perf_counter() - t) / 5 * 1e3, 2)
    assert (ctx.decrypt(sk, r) == ~(m & ~m)).all()
print(engine, out)
