"""NAND timing with tlwe_mask_size=2 (4096 bits and a small batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch, nufhe_amd
ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(1))
sk, ck = ctx.make_key_pair(tlwe_mask_size=2)
vm = ctx.make_virtual_machine(ck)
for B in (32, 1536, 4096):
    m = numpy.random.RandomState(1).randint(0, 2, size=B).astype(bool)
    c1 = ctx.encrypt(sk, m); c2 = ctx.encrypt(sk, ~m)
    r = vm.gate_nand(c1, c2); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3): r = vm.gate_nand(c1, c2)
    torch.cuda.synchronize()
    print("k=2 NAND %d bits: %.2f ms, correct=%s" % (B, (time.time() - t) / 3 * 1e3, bool((ctx.decrypt(sk, r) == ~(m & ~m)).all())))
