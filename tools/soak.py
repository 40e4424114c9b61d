"""Soak test: repeats gates many times and requires bit-identical ciphertexts every time
(the kernels synchronise waves with fences only / work-group barriers in the team kernel / LDS arrival
counters in the pair kernel; any race
would show up as a run-to-run difference) and correct decryptions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch, nufhe_amd

def soak(ctx, vm, sk, B, reps, gate):
    rs = numpy.random.RandomState(B)
    m = [rs.randint(0, 2, size=B).astype(bool) for _ in range(3)]
    c = [ctx.encrypt(sk, x) for x in m]
    fn = (lambda: vm.gate_mux(c[0], c[1], c[2])) if gate == 'mux' else (lambda: vm.gate_nand(c[0], c[1]))
    ref = fn()
    truth = numpy.where(m[0], m[1], m[2]) if gate == 'mux' else ~(m[0] & m[1])
    assert (ctx.decrypt(sk, ref) == truth).all()
    bad = 0
    t = time.time()
    for _ in range(reps):
        r = fn()
        same = bool(torch.equal(r.a, ref.a) and torch.equal(r.b, ref.b) and torch.equal(r.current_variances, ref.current_variances))
        bad += not same
    print("%s %5d bits x %4d: %d differing runs, %.1f s" % (gate, B, reps, bad, time.time() - t), flush=True)
    return bad

def main():
    total = 0
    # `python tools/soak.py exact-fft`: the NTT key on the exact-FFT engine only (team barriers through arrival words and
    # s_barrier, LDS atomics into the accumulator, parked accumulators in global memory)
    engine = sys.argv[1] if len(sys.argv) > 1 else None
    for tr in (('NTT',) if engine else ('NTT', 'FFT')):
        ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(5))
        sk, ck = ctx.make_key_pair(transform_type=tr)
        if engine:
            ck.set_engine(engine)
        vm = ctx.make_virtual_machine(ck)
        print(tr, engine or '')
        for B, reps in ((4096, 60 if tr == 'NTT' else 200), (2500, 60), (1000, 100), (400, 150), (200, 300), (7, 300)) + (((512, 150), (300, 150)) if engine else ()):
            total += soak(ctx, vm, sk, B, reps, 'nand')
        total += soak(ctx, vm, sk, 300, 100, 'mux')
    print("TOTAL differing runs:", total)
    sys.exit(1 if total else 0)

if __name__ == '__main__':
    main()
