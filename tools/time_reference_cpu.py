"""
time_reference_cpu.py -- SURVEY §8(d) CPU baseline (i): the REFERENCE's Python CPU path.

Times K in {2, 3, 4, 5} blind-rotate iterations of the composition of the reference's own CPU
functions (nufhe/polynomials_cpu.py shift + nufhe/tgsw_cpu.py external product, in the order of
nufhe/bootstrap.py:96-142) at batch B = 32 on ONE core, fits t(K) = t0 + K * t_iter and reports the
500-iteration extrapolation, labelled as such.  The reference exists only in the build container
(/root/reference), not on the GPU box, so this number is taken on the container CPU and committed
under profiles/; bench.py's `cpu_baseline` (the C oracle, all host cores) is the one measured on the
GPU box in the same run.  Test/measurement infrastructure only.
"""

import json
import os
import platform
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

from oracle import ref_shim          # noqa: E402
from oracle import oracle as orc     # noqa: E402


def main():
    B = 32
    ref = ref_shim.load()
    tl = ref_shim.RefTLweParams(1024, 1)
    tg = ref_shim.RefTGswParams(tl, 2, 10)
    lwe_key, tlwe_key, ck = orc.make_key_pair(orc.DeterministicRNG(123))
    rng = orc.DeterministicRNG(456)
    m1 = rng.uniform_bool((B,)).astype(bool); m2 = rng.uniform_bool((B,)).astype(bool)
    c1 = orc.encrypt(rng, lwe_key, m1); c2 = orc.encrypt(rng, lwe_key, m2)
    MU = numpy.int32(2**29)
    ta = (-c1[0] - c2[0]).astype(numpy.int32)
    tb = (MU - c1[1] - c2[1]).astype(numpy.int32)
    N = 1024
    shape = (B,)
    barb = numpy.empty(shape, numpy.int32); bara = numpy.empty(ta.shape, numpy.int32)
    ref.numeric_functions_cpu.Torus32ToPhaseReference(shape, 2 * N)(barb, tb)
    ref.numeric_functions_cpu.Torus32ToPhaseReference(ta.shape, 2 * N)(bara, ta)
    testvect = numpy.full(shape + (N,), MU, numpy.int32)
    tvb = numpy.empty_like(testvect)
    ref.polynomials_cpu.ShiftTorusPolynomialReference(N, shape, shape, invert_powers=True)(tvb, testvect, barb, 0)
    acc0 = numpy.empty(shape + (2, N), numpy.int32); cv = numpy.empty(shape, numpy.float32)
    ref.tlwe_cpu.TLweNoiselessTrivialReference(tl, shape)(acc0, cv, tvb)
    shift = ref.polynomials_cpu.ShiftTorusPolynomialReference(N, acc0.shape[:-1], bara.shape, powers_view=True, minus_one=True)
    extmul = ref.tgsw_cpu.TGswTransformedExternalMulReference(tg, shape, ck.bk.shape[0], None)

    def run(K):
        acc = acc0.copy()
        t0 = time.perf_counter()
        for i in range(K):
            tmp = numpy.empty_like(acc)
            shift(tmp, acc, bara, i)
            extmul(tmp, ck.bk, i)
            acc = acc + tmp
        return time.perf_counter() - t0, acc

    run(1)
    Ks = [2, 3, 4, 5]
    ts = []
    for K in Ks:
        t, acc = run(K)
        ts.append(t)
        print("K=%d: %.2f s" % (K, t), flush=True)
    # parity of the timed composition with the oracle (5 iterations)
    exp = orc.blind_rotate(acc0, ck.bk, bara, n_iter=5)
    assert (exp == acc).all()
    t_iter, t0 = numpy.polyfit(Ks, ts, 1)
    cpu = platform.processor() or ""
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                cpu = line.split(':', 1)[1].strip(); break
    except OSError:
        pass
    out = {
        "what": "reference Python CPU functions (polynomials_cpu shift + tgsw_cpu external product), B=32, 1 core",
        "cpu": cpu, "cores_used": 1, "K": Ks, "seconds": ts,
        "seconds_per_iteration_B32": float(t_iter),
        "extrapolated_blind_rotate_500_iterations_s": float(500 * t_iter),
        "extrapolated_ms_per_bit": float(500 * t_iter / B * 1e3),
        "label": "EXTRAPOLATED x(500/K) from K<=5 iterations; excludes keyswitch (<1% of the gate)",
    }
    path = os.path.join(ROOT, 'profiles', 'reference_python_cpu_timing.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
