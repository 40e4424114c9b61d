"""NAND timing with tlwe_mask_size=2, both transforms, batch sizes that exercise every k = 2 kernel (team kernels up to
2 x CUs bits, the two builds of the NTT wave kernel, the FFT wave kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy, torch, nufhe_amd
from nufhe_amd import _lib
# `python tools/time_k2.py NTT:2 NTT:1` = the NTT path with the quad kernel (ring_k2 = 2), then with the ring kernel;
# `NTT:x` = the NTT key on the exact-fft engine
runs = [(a.split(':') + [None])[:2] for a in sys.argv[1:]] or [('FFT', None), ('NTT', None)]
for tr, ring in runs:
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(1))
    exact = ring == 'x'
    if exact:
        ring = None
    if ring is not None:
        t = _lib.NufheTuning()
        _lib.check(_lib.lib().nufhe_ctx_get_tuning(ctx.thread.handle, ctypes.byref(t)))
        t.ring_k2 = int(ring)
        _lib.check(_lib.lib().nufhe_ctx_set_tuning(ctx.thread.handle, ctypes.byref(t)))
        print("ring_k2 = %s" % ring)
    sk, ck = ctx.make_key_pair(tlwe_mask_size=2, transform_type=tr)
    if exact:
        ck.set_engine('exact-fft')
        print('engine exact-fft')
    vm = ctx.make_virtual_machine(ck)
    for B in (32, 256, 512, 768, 1024, 1536, 2048, 3000, 4096):
        m = numpy.random.RandomState(1).randint(0, 2, size=B).astype(bool)
        c1 = ctx.encrypt(sk, m); c2 = ctx.encrypt(sk, ~m)
        r = vm.gate_nand(c1, c2); torch.cuda.synchronize()
        t = time.time()
        for _ in range(3): r = vm.gate_nand(c1, c2)
        torch.cuda.synchronize()
        print("%s k=2 NAND %d bits: %.2f ms, correct=%s" % (tr, B, (time.time() - t) / 3 * 1e3, bool((ctx.decrypt(sk, r) == ~(m & ~m)).all())), flush=True)
