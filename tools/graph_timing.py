"""graph_timing.py -- uint_min (4 x 16 and 128 x 32 bits) and a 16-gate NAND chain on 4 bits: eager calls against ONE replayed
hipGraph (nufhe_amd/graph.py).  Median of 5 synchronised runs each.  Prints one JSON line (profiles/r04_graph_capture.json)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
import nufhe_amd as nufhe
from nufhe_amd.device import DeviceThread
from nufhe_amd.operators_integer import uint_min, uintarray_to_bitarray

stream = torch.cuda.Stream()
out = {}
with torch.cuda.stream(stream):
    thr = DeviceThread(0)
    ctx = nufhe.Context(rng=nufhe.DeterministicRNG(123), thread=thr)
    secret, cloud = ctx.make_key_pair()
    vm = ctx.make_virtual_machine(cloud)
    rs = numpy.random.RandomState(3)

    def timed(fn, n=5):
        fn(); thr.synchronize()
        ts = []
        for _ in range(n):
            t = time.perf_counter(); fn(); thr.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
        return round(sorted(ts)[n // 2], 3)

    for shape, dt in (((4,), numpy.uint16), ((128,), numpy.uint32)):
        xs = [rs.randint(0, 2**16, size=shape).astype(dt) for _ in range(2)]
        a = ctx.encrypt(secret, uintarray_to_bitarray(xs[0])); b = ctx.encrypt(secret, uintarray_to_bitarray(xs[1]))
        answer = nufhe.api_low_level.empty_ciphertext(thr, cloud.params, a.shape)
        g = nufhe.GateGraph(thr)
        g.capture(lambda: uint_min(thr, cloud, answer, a, b))
        key = "uint_min_%dx%d" % (shape[0], a.shape[-1])
        out[key] = {"eager_ms": timed(lambda: uint_min(thr, cloud, answer, a, b)), "graph_replay_ms": timed(g.replay),
                    "gates": 2 + a.shape[-1]}
    c = [ctx.encrypt(secret, rs.randint(0, 2, 4).astype(bool)) for _ in range(2)]

    def chain():
        t = c[0]
        for _ in range(16):
            t = vm.gate_nand(t, c[1])
        return t
    g = nufhe.GateGraph(thr)
    g.capture(chain)
    out["nand_chain_16_gates_4_bits"] = {"eager_ms": timed(chain), "graph_replay_ms": timed(g.replay), "gates": 16}
print(json.dumps(out))
