"""fuzz_engines.py [iterations] [seed] -- differential fuzz of the two engines of an NTT key: random gates (binary, MUX) on
random batch sizes (biased to the dispatch boundaries: 1, 2, 4, 8 and 10 x CUs +- a few bits) and random views (offset
slices), tlwe_mask_size 1 and 2: every output word of the exact-FFT engine must equal the native engine's.  Prints one JSON line.
`fuzz_engines.py N SEED fft`: an FFT key made from the same seed (same secret key, same key polynomials) against the NTT key's
native engine instead: the FFT path's stated tolerance (16 LSB per word), and how many words differ at all (observed: 0)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
import nufhe_amd as nufhe

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
fft_mode = len(sys.argv) > 3 and sys.argv[3] == 'fft'
rs = numpy.random.RandomState(seed)
cus = torch.cuda.get_device_properties(0).multi_processor_count
BIN = ['gate_nand', 'gate_or', 'gate_and', 'gate_xor', 'gate_xnor', 'gate_nor', 'gate_andny', 'gate_andyn', 'gate_orny', 'gate_oryn']
res = {"mode": "fft key vs ntt key" if fft_mode else "exact-fft vs native engine", "iterations": 0, "words": 0, "differing": 0,
       "wrong_bits": 0, "max_abs_diff_lsb": 0, "sizes": []}
t0 = time.time()
for k in (1, 2):
    ctx = nufhe.Context(rng=nufhe.DeterministicRNG(seed + k))
    sk, ck = ctx.make_key_pair(tlwe_mask_size=k)
    vm = ctx.make_virtual_machine(ck)
    if fft_mode:
        ctx_f = nufhe.Context(rng=nufhe.DeterministicRNG(seed + k), thread=ctx.thread)
        sk_f, ck_f = ctx_f.make_key_pair(tlwe_mask_size=k, transform_type='FFT')
        vm_f = ctx_f.make_virtual_machine(ck_f)
    top = (10 if k == 1 else 5) * cus + 40
    ms = [rs.randint(0, 2, size=top).astype(bool) for _ in range(3)]
    cs = [ctx.encrypt(sk, m) for m in ms]
    n = iters if k == 1 else iters // 3
    for it in range(n):
        if rs.rand() < 0.6:
            B = int(max(1, rs.choice([1, 2, 4, 8, 10][:5 if k == 1 else 3]) * cus + rs.randint(-3, 4)))
        else:
            B = int(rs.randint(1, top - 8))
        B = min(B, top - 8)
        off = int(rs.randint(0, 8))
        a, b, c = (x[off:off + B] for x in cs)
        name = 'gate_mux' if rs.rand() < 0.3 else BIN[rs.randint(len(BIN))]
        out = {}
        for engine in ('native', 'exact-fft'):
            if fft_mode:
                v = vm if engine == 'native' else vm_f
            else:
                ck.set_engine(engine)
                v = vm
            r = v.gate_mux(a, b, c) if name == 'gate_mux' else getattr(v, name)(a, b)
            out[engine] = (r.a.clone(), r.b.clone(), r.current_variances.clone(), r)
        bad = sum(int((x != y).sum()) for x, y in zip(out['native'][:3], out['exact-fft'][:3]))
        if fft_mode and bad:
            for x, y in zip(out['native'][:2], out['exact-fft'][:2]):
                d = ((x.to(torch.int64) - y.to(torch.int64) + 2**31) % 2**32 - 2**31).abs().max().item()
                res["max_abs_diff_lsb"] = max(res["max_abs_diff_lsb"], int(d))
        res["differing"] += bad
        res["words"] += int(out['native'][0].numel() + 2 * out['native'][1].numel())
        if name in ('gate_nand', 'gate_mux'):
            truth = numpy.where(ms[0][off:off + B], ms[1][off:off + B], ms[2][off:off + B]) if name == 'gate_mux' else ~(ms[0][off:off + B] & ms[1][off:off + B])
            res["wrong_bits"] += int((ctx.decrypt(sk, out['exact-fft'][3]) != truth).sum())
        res["iterations"] += 1
        if bad:
            res["sizes"].append([k, name, B, off, bad])
res["seconds"] = round(time.time() - t0, 1)
print(json.dumps(res))
sys.exit(1 if res["wrong_bits"] or (res["max_abs_diff_lsb"] > 16 if fft_mode else res["differing"]) else 0)
