"""20 NAND gates each on 1, 256 and 512 bits (engine = argv[1], default exact-fft): the workload of
`rocprofv3 --kernel-trace --stats -- python tools/small_gates_for_profiling.py` -> profiles/r06j_small_batch_kernel_stats.csv"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch, nufhe_amd
ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(1))
sk, ck = ctx.make_key_pair()
ck.set_engine(sys.argv[1] if len(sys.argv) > 1 else 'exact-fft')
vm = ctx.make_virtual_machine(ck)
for B in (1, 256, 512):
    m = numpy.random.RandomState(B).randint(0, 2, size=B).astype(bool)
    a = ctx.encrypt(sk, m); b = ctx.encrypt(sk, ~m)
    for _ in range(20):
        r = vm.gate_nand(a, b)
    torch.cuda.synchronize()
