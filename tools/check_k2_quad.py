"""check_k2_quad.py -- k = 2 NTT: the quad kernel (ring_k2 = 2), the ring kernel (1) and the wave kernels (0) must write
identical ciphertext words (NAND and MUX, sizes around the team limit and a ragged last work-group)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch, nufhe_amd
from nufhe_amd import _lib

ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(1))
sk, ck = ctx.make_key_pair(tlwe_mask_size=2, transform_type='NTT')
vm = ctx.make_virtual_machine(ck)


def set_ring(v):
    t = _lib.NufheTuning()
    _lib.check(_lib.lib().nufhe_ctx_get_tuning(ctx.thread.handle, ctypes.byref(t)))
    t.ring_k2 = v
    _lib.check(_lib.lib().nufhe_ctx_set_tuning(ctx.thread.handle, ctypes.byref(t)))


rs = numpy.random.RandomState(3)
ok = True
for B in (257, 301, 777):
    ms = [rs.randint(0, 2, size=B).astype(bool) for _ in range(3)]
    cs = [ctx.encrypt(sk, m) for m in ms]
    res = {}
    for v in (2, 1, 0):
        set_ring(v)
        res[v] = (vm.gate_nand(cs[0], cs[1]), vm.gate_mux(cs[0], cs[1], cs[2]))
    same = all(res[2][i] == res[1][i] and res[2][i] == res[0][i] for i in range(2))
    dec = bool((ctx.decrypt(sk, res[2][0]) == ~(ms[0] & ms[1])).all() and
               (ctx.decrypt(sk, res[2][1]) == numpy.where(ms[0], ms[1], ms[2])).all())
    print("k=2 NTT %d bits: quad == ring == wave: %s, decrypts: %s" % (B, same, dec), flush=True)
    ok = ok and same and dec
sys.exit(0 if ok else 1)
