"""stage_times.py -- HIP-event time of the bootstrap and keyswitch stages of a NAND (nufhe_profile_last) and the wall time
of the call, by batch size.  Prints one JSON line."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import nufhe_amd as nufhe
from nufhe_amd import _lib

ctx = nufhe.Context(rng=nufhe.DeterministicRNG(123))
secret, cloud = ctx.make_key_pair()
vm = ctx.make_virtual_machine(cloud)
lib = _lib.lib()
rs = numpy.random.RandomState(3)
lib.nufhe_profile_enable(ctx.thread.handle, 1)
out = {}
for bits in (1, 64, 256, 512, 1024, 4096):
    cts = [ctx.encrypt(secret, rs.randint(0, 2, bits).astype(bool)) for _ in range(2)]
    vm.gate_nand(cts[0], cts[1]); ctx.thread.synchronize()
    rows = []
    for _ in range(5):
        t = time.perf_counter(); vm.gate_nand(cts[0], cts[1]); ctx.thread.synchronize(); wall = (time.perf_counter() - t) * 1e3
        a = ctypes.c_float(); b = ctypes.c_float()
        lib.nufhe_profile_last(ctx.thread.handle, ctypes.byref(a), ctypes.byref(b))
        rows.append((wall, a.value, b.value))
    rows.sort()
    w, br, ks = rows[2]
    out[bits] = {"wall_ms": round(w, 3), "bootstrap_ms": round(br, 3), "keyswitch_ms": round(ks, 3), "rest_ms": round(w - br - ks, 3)}
print(json.dumps(out))
