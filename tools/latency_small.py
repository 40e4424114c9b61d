"""latency_small.py -- NAND / MUX gate time on the smallest batches (the 8- and 4-waves-per-bit NTT kernels):
median of 7 timed calls each.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import nufhe_amd as nufhe
from nufhe_amd import _lib

ctx = nufhe.Context(rng=nufhe.DeterministicRNG(123))
secret, cloud = ctx.make_key_pair()
vm = ctx.make_virtual_machine(cloud)
rs = numpy.random.RandomState(3)
out = {}
for team8 in (1, 0):
    _lib.call("nufhe_ctx_set_team8", ctx.thread.handle, team8)
    for bits in (1, 64, 256):
        cts = [ctx.encrypt(secret, rs.randint(0, 2, bits).astype(bool)) for _ in range(3)]
        for name, call in (('nand', lambda: vm.gate_nand(cts[0], cts[1])), ('mux', lambda: vm.gate_mux(*cts))):
            if name == 'mux' and bits > 128:
                continue
            call(); ctx.thread.synchronize()
            ts = []
            for _ in range(7):
                t = time.perf_counter(); call(); ctx.thread.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
            out['%s_%d_team%d' % (name, bits, 8 if team8 else 4)] = round(sorted(ts)[3], 3)
print(json.dumps(out))
