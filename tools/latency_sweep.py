"""Gate time vs batch size (small batches = circuit-style use) and the uint_min chain of the
reference's perf test (test/test_gates.py:317-353).  Prints one JSON object."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy
import torch

import nufhe_amd
from nufhe_amd.operators_integer import uint_min, uintarray_to_bitarray, bitarray_to_uintarray


def main():
    transform = sys.argv[1] if len(sys.argv) > 1 else 'NTT'
    engine = sys.argv[2] if len(sys.argv) > 2 else 'native'           # 'exact-fft': the fp64 engine of NTT keys
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(123))
    sk, ck = ctx.make_key_pair(transform_type=transform)
    if engine != 'native':
        ck.set_engine(engine)
    vm = ctx.make_virtual_machine(ck)
    rs = numpy.random.RandomState(1)
    out = {"transform": transform, "engine": engine, "nand_ms": {}, "mux_ms": {}}
    for B in (1, 32, 128, 256, 512, 1024, 2048, 3000, 4096, 8192):
        m = [rs.randint(0, 2, size=B).astype(bool) for _ in range(3)]
        c = [ctx.encrypt(sk, x) for x in m]
        d = vm.empty_ciphertext((B,))
        for name, fn in (("nand_ms", lambda: vm.gate_nand(c[0], c[1], dest=d)),
                         ("mux_ms", lambda: vm.gate_mux(c[0], c[1], c[2], dest=d))):
            fn(); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            out[name][B] = (time.perf_counter() - t) / 3 * 1e3
    for shape in ((4, 16), (128, 32)):
        dt = {16: numpy.uint16, 32: numpy.uint32}[shape[1]]
        x = rs.randint(0, 2**shape[1], size=shape[0]).astype(dt); y = rs.randint(0, 2**shape[1], size=shape[0]).astype(dt)
        ca = ctx.encrypt(sk, uintarray_to_bitarray(x)); cb = ctx.encrypt(sk, uintarray_to_bitarray(y))
        ans = nufhe_amd.empty_ciphertext(ctx.thread, ck.params, shape)
        uint_min(ctx.thread, ck, ans, ca, cb); torch.cuda.synchronize()
        t = time.perf_counter()
        uint_min(ctx.thread, ck, ans, ca, cb); torch.cuda.synchronize()
        el = time.perf_counter() - t
        ok = bool((bitarray_to_uintarray(ctx.decrypt(sk, ans)) == numpy.minimum(x, y)).all())
        out["uint_min_%dx%d" % shape] = {"ms": el * 1e3, "ms_per_bit": el * 1e3 / (shape[0] * shape[1]), "correct": ok}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
