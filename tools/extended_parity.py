"""One-off extended parity run (not part of the test-suite: ~2 minutes of CPU oracle time on the GPU
box): every output word of 2048-bit NAND / XOR / MUX gates from the GPU (wave-per-bit kernel), of
700-bit gates (2-waves-per-bit kernel), 200-bit gates (8-waves-per-bit half-ring kernel; the MUX = 400 bootstraps goes
to the pair kernel) and of a 256-bit k=2 NAND equals the CPU oracle.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy
import nufhe_amd
import gpu_helpers as H
from nufhe_amd.device import DeviceThread
from oracle import oracle as orc


def run(thr, params, oparams, seed, sizes):
    lwe_key, tlwe_key, ck = orc.make_key_pair(orc.DeterministicRNG(seed), oparams)
    cloud = H.cloud_key_from_arrays(thr, ck, params)
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(1), thread=thr)
    vm = ctx.make_virtual_machine(cloud)
    out = {}
    for B, gates in sizes:
        rng = orc.DeterministicRNG(1000 + B)
        ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(3)]
        cs = [orc.encrypt(rng, lwe_key, m, oparams) for m in ms]
        ds = [H.ciphertext_from_arrays(thr, c, params) for c in cs]
        for g in gates:
            t = time.time()
            if g == 'mux':
                exp = orc.gate_mux(ck, cs[0], cs[1], cs[2]); got = vm.gate_mux(ds[0], ds[1], ds[2])
            else:
                exp = orc.gate('gate_' + g, ck, cs[0], cs[1]); got = getattr(vm, 'gate_' + g)(ds[0], ds[1])
            ra, rb, rcv = H.ct_arrays(got)
            bad = int((ra != exp[0]).sum() + (rb != exp[1]).sum() + (rcv != exp[2]).sum())
            out["%s_%d" % (g, B)] = {"words": int(ra.size + rb.size + rcv.size), "differing": bad,
                                     "oracle_s": round(time.time() - t, 1)}
            print(g, B, out["%s_%d" % (g, B)], flush=True, file=sys.stderr)
    return out


def main():
    # `NUFHE_NTT_ENGINE=exact-fft python tools/extended_parity.py`: the same legs on the exact-FFT engine (every new NTT key
    # takes the engine of that variable: the kernels are then k_bootstrap_xfft, _quad (<= 2 x CUs bits), _k2 / _hex_k2)
    thr = DeviceThread(0)
    res = {"engine": os.environ.get("NUFHE_NTT_ENGINE", "native"),
           "k1": run(thr, nufhe_amd.NuFHEParameters(), orc.Params(), 123, [(2048, ['nand', 'xor', 'mux']), (700, ['nand']), (200, ['nand', 'xnor', 'mux']), (100, ['mux'])]),
           "k2": run(thr, nufhe_amd.NuFHEParameters(tlwe_mask_size=2), orc.Params(mask_size=2), 123, [(256, ['nand']), (600, ['nand'])])}
    res["total_differing"] = sum(v["differing"] for part in ("k1", "k2") for v in res[part].values())
    print(json.dumps(res))
    sys.exit(1 if res["total_differing"] else 0)


if __name__ == '__main__':
    main()
