"""NAND latency of the fp64 kernel families (exact engine and FFT keys, k = 1 and 2) at 1 / 256 / 512 / 4096 bits; one dict."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch, nufhe_amd
res = {}
for tr, eng, k in (('NTT', 'exact-fft', 1), ('FFT', None, 1), ('NTT', 'exact-fft', 2), ('FFT', None, 2)):
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(1))
    sk, ck = ctx.make_key_pair(transform_type=tr, tlwe_mask_size=k)
    if eng: ck.set_engine(eng)
    vm = ctx.make_virtual_machine(ck)
    for B in (1, 256, 512, 4096):
        m = numpy.random.RandomState(B).randint(0, 2, size=B).astype(bool)
        a = ctx.encrypt(sk, m); b = ctx.encrypt(sk, ~m)
        r = vm.gate_nand(a, b); torch.cuda.synchronize()
        n = 10 if B <= 512 else 4
        t = time.perf_counter()
        for _ in range(n): r = vm.gate_nand(a, b)
        torch.cuda.synchronize()
        res["%s%s_k%d_%d" % (tr, '_x' if eng else '', k, B)] = round((time.perf_counter() - t) / n * 1e3, 3)
print(res)
