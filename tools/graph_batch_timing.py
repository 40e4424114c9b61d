"""graph_batch_timing.py -- four independent uint_min circuits in lock step (18 gate_batch calls): eager against ONE replayed
hipGraph (nufhe_amd.GateGraph; gate_batch is capturable because its job tables travel inside kernel arguments).
Prints one JSON line."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
import nufhe_amd as nufhe
from nufhe_amd.device import DeviceThread
from nufhe_amd.operators_integer import uint_min_many, uintarray_to_bitarray
stream = torch.cuda.Stream()
rs = numpy.random.RandomState(8)
with torch.cuda.stream(stream):
    thr = DeviceThread(0)
    ctx = nufhe.Context(rng=nufhe.DeterministicRNG(43), thread=thr)
    secret, cloud = ctx.make_key_pair()
    engine = sys.argv[1] if len(sys.argv) > 1 else 'native'       # `python tools/graph_batch_timing.py exact-fft`
    cloud.set_engine(engine)
    C, M, W = 4, 4, 16
    ca = [ctx.encrypt(secret, uintarray_to_bitarray(rs.randint(0, 2**W, M).astype(numpy.uint16))) for _ in range(C)]
    cb = [ctx.encrypt(secret, uintarray_to_bitarray(rs.randint(0, 2**W, M).astype(numpy.uint16))) for _ in range(C)]
    outs = [nufhe.empty_ciphertext(thr, cloud.params, (M, W)) for _ in range(C)]
    def circuit():
        uint_min_many(thr, cloud, outs, ca, cb)
        return outs
    def timed(fn, reps=5):
        fn(); thr.synchronize()
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); fn(); thr.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
        return round(sorted(ts)[reps // 2], 3)
    eager = timed(circuit)
    g = nufhe.GateGraph(thr)
    g.capture(circuit)
    replay = timed(g.replay)
    print(json.dumps({"engine": engine, "four_uint_min_4x16_lock_step": {"eager_ms": eager, "one_graph_replay_ms": replay, "batches": W + 2}}))
