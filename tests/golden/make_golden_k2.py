"""
make_golden_k2.py -- golden vectors for tlwe_mask_size = 2 (the reference's second tested
parameter set, test/test_gates.py:96-100), produced by the REFERENCE's own CPU
functions (loaded through oracle/ref_shim.py):

  * TGswTransformedExternalMul with k = 2 (tgsw_cpu.py:82-106), small and full-range accumulators
  * TLweEncryptZero with k = 2 (tlwe_cpu.py:64-89)
  * a complete NAND gate, batch 2, n = 500, k = 2: gates.py:110-121 -> bootstrap.py:206-229 -> 119-142
    -> keyswitch from LWE(2048)

Keys: oracle.make_key_pair(DeterministicRNG(123), Params(mask_size=2)); ciphertexts from
DeterministicRNG(456).  Output: tests/golden/reference_outputs_k2.npz.  Slow (minutes).
"""

import os
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import ref_shim          # noqa: E402
from oracle import oracle as orc     # noqa: E402
import golden_inputs as gi           # noqa: E402
from make_golden_gate import ref_bootstrap   # noqa: E402


def gate_inputs_k2():
    params = orc.Params(mask_size=2)
    lwe_key, tlwe_key, ck = orc.make_key_pair(orc.DeterministicRNG(123), params)
    rng = orc.DeterministicRNG(456)
    ms = [numpy.array([True, False]), numpy.array([True, True])]
    cts = [orc.encrypt(rng, lwe_key, m, params) for m in ms]
    return lwe_key, tlwe_key, ck, cts, ms


def main():
    ref = ref_shim.load()
    tl = ref_shim.RefTLweParams(1024, 2)
    tg = ref_shim.RefTGswParams(tl, 2, 10)
    out = {}

    for full in (False, True):
        accum, bk, row = gi.extmul_inputs_k2(full_range=full)
        acc = accum.copy()
        ref.tgsw_cpu.TGswTransformedExternalMulReference(tg, accum.shape[:-2], bk.shape[0], None)(acc, bk, row)
        out['tgsw_extmul_k2' + ('_full' if full else '')] = acc

    key, n1, n2 = gi.encrypt_zero_inputs_k2()
    shape = n2.shape[:-1]
    ra = numpy.empty(shape + (3, 1024), numpy.int32); rcv = numpy.empty(shape, numpy.float32)
    ref.tlwe_cpu.TLweEncryptZeroReference(tl, shape, 9e-9, None)(ra, rcv, key, n1, n2)
    out['encrypt_zero_k2_a'] = ra; out['encrypt_zero_k2_cv'] = rcv

    lwe_key, tlwe_key, ck, cts, ms = gate_inputs_k2()
    MU = numpy.int32(2**29)
    ta = (-cts[0][0] - cts[1][0]).astype(numpy.int32)
    tb = (MU - cts[0][1] - cts[1][1]).astype(numpy.int32)
    ea, eb = ref_bootstrap(ref, tl, tg, ck.bk, ta, tb, MU)
    out['nand_k2_ext_a'] = ea; out['nand_k2_ext_b'] = eb
    ks = ref.lwe_cpu.LweKeyswitchReference(None, 2048, 500, 8, 2)
    ra = numpy.empty((2, 500), numpy.int32); rb = numpy.empty((2,), numpy.int32); rcv = numpy.empty((2,), numpy.float32)
    ks(ra, rb, rcv, ck.ks_a, ck.ks_b, ck.ks_cv, ea, eb)
    out['nand_k2_a'] = ra; out['nand_k2_b'] = rb; out['nand_k2_cv'] = rcv
    assert ((orc.lwe_decrypt(ra, rb, lwe_key) > 0) == ~(ms[0] & ms[1])).all()

    path = os.path.join(HERE, 'reference_outputs_k2.npz')
    numpy.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == '__main__':
    main()
