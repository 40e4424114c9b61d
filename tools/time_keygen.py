"""Cloud-key generation timing: Context.make_key_pair for every (transform, mask size),
second call of each (no one-time costs), with the host random-number share measured separately."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch, nufhe_amd

ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(1))
for tr, k in (("NTT", 1), ("FFT", 1), ("NTT", 2), ("FFT", 2)):
    for rep in range(2):
        ctx.rng = nufhe_amd.DeterministicRNG(1)
        torch.cuda.synchronize(); t = time.time()
        sk, ck = ctx.make_key_pair(transform_type=tr, tlwe_mask_size=k)
        torch.cuda.synchronize(); dt = time.time() - t
        del sk, ck
    rng = nufhe_amd.DeterministicRNG(1)
    t = time.time()
    rng.uniform_bool((500,)); rng.uniform_bool((k, 1024))
    rng.uniform_torus32((500, k + 1, 2, k, 1024)); rng.gauss((500, k + 1, 2, 1024), 1e-8)
    rng.gauss((1024 * k, 8, 3), 1e-5); rng.uniform_torus32((1024 * k, 8, 3, 500))
    t_rng = time.time() - t
    print("keygen %s k=%d: %.1f ms (host random numbers alone: %.1f ms)" % (tr, k, dt * 1e3, t_rng * 1e3))
