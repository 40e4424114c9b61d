"""
make_golden.py -- generate golden vectors by executing the REFERENCE's own CPU reference
functions (nucypher/nufhe ``*_cpu.py``, loaded through oracle/ref_shim.py) on seeded inputs.

Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

Inputs are NOT stored: every case regenerates them from a ``numpy.random.RandomState(seed)``
(the legacy generator is bit-stable across NumPy versions) through ``golden_inputs.py``, which the
tests import as well.  Only the reference's outputs are written to ``tests/golden/*.npz``.
Value ranges and shapes mirror the reference's own differential tests (cited per case).
"""

import os
import sys
import time

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import ref_shim  # noqa: E402
import golden_inputs as gi   # noqa: E402


def main():
    ref = ref_shim.load()
    g = ref.ntt_cpu
    out = {}
    t0 = time.time()

    # ---- finite field arithmetic (test/test_transform/test_arithmetic.py) ----
    a, b, e, s = gi.ff_inputs()
    P = g.GaloisNumber.modulus
    ga, gb = g.gnum(a), g.gnum(b)
    out['ff_add'] = g.gnum_to_u64(ga + gb)
    out['ff_sub'] = g.gnum_to_u64(ga - gb)
    out['ff_mul'] = g.gnum_to_u64(ga * gb)
    out['ff_mul_prepared'] = ref.transform.transformed_space_mul_prepared_ref(a, b)
    out['ff_pow'] = numpy.array([(g.GaloisNumber(int(x))**int(y)).val for x, y in zip(a, e)], numpy.uint64)
    out['ff_lsh'] = numpy.array([(int(x) * 2**int(y)) % P for x, y in zip(a, s)], numpy.uint64)
    out['ff_to_i32'] = g.gnum_to_i32(ga)
    # KATs the reference's tests hold
    out['kat_root_64'] = numpy.uint64((g.GaloisNumber(0xa70dc47e4cbdf43f)**(2**32 // 64)).val)
    out['kat_rinv'] = numpy.uint64(g.GaloisNumber(2**64).inverse().val)      # test_arithmetic.py:178
    out['kat_r'] = numpy.uint64(g.GaloisNumber(2**64).val)                   # test_arithmetic.py:193

    # ---- transforms (test/test_transform/test_computation.py:33-68) ----
    polys_i32, polys_ff = gi.ntt_inputs()
    out['ntt_forward_i32'] = ref.transform.ntt_transform_ref(polys_i32, i32_conversion=True)
    out['ntt_forward_u64'] = ref.transform.ntt_transform_ref(polys_ff, i32_conversion=False)
    out['ntt_inverse_i32'] = ref.transform.ntt_transform_ref(polys_ff, inverse=True, i32_conversion=True)
    out['ntt_inverse_u64'] = ref.transform.ntt_transform_ref(polys_ff, inverse=True, i32_conversion=False)
    small = gi.ntt_small_inputs()
    out['ntt_small_forward'] = ref.transform.ntt_transform_ref(small, i32_conversion=True)
    print("transforms done", time.time() - t0)

    # ---- mod-switch (test/test_numeric_functions.py:27-45) ----
    phase = gi.modswitch_inputs()
    res = numpy.empty(phase.shape, numpy.int32)
    ref.numeric_functions_cpu.Torus32ToPhaseReference(phase.shape, 2048)(res, phase)
    out['t32_to_phase'] = res

    # ---- shift (test/test_polynomials.py:30-58), N=16 and N=1024 ----
    for tag, (src, powers, N) in gi.shift_inputs().items():
        for minus_one in (False, True):
            for invert in (False, True):
                r = numpy.empty_like(src)
                ref.polynomials_cpu.ShiftTorusPolynomialReference(
                    N, src.shape[:-1], powers.shape, powers_view=False,
                    minus_one=minus_one, invert_powers=invert)(r, src, powers, 0)
                out['shift_%s_m%d_i%d' % (tag, minus_one, invert)] = r
    # powers_view variant (per-iteration use, tlwe.py:168-169)
    src, powers_arr, idx, N = gi.shift_view_inputs()
    r = numpy.empty_like(src)
    ref.polynomials_cpu.ShiftTorusPolynomialReference(
        N, src.shape[:-1], powers_arr.shape, powers_view=True, minus_one=True)(r, src, powers_arr, idx)
    out['shift_view'] = r

    # ---- TLWE trivial / extract (test/test_tlwe.py:38-94) ----
    tl = ref_shim.RefTLweParams(1024, 1)
    mu = gi.tlwe_trivial_inputs()
    shape = mu.shape[:-1]
    a_ = numpy.empty(shape + (2, 1024), numpy.int32); cv = numpy.empty(shape, numpy.float32)
    ref.tlwe_cpu.TLweNoiselessTrivialReference(tl, shape)(a_, cv, mu)
    out['tlwe_trivial_a'] = a_
    tl_a = gi.tlwe_extract_inputs()
    shape = tl_a.shape[:-2]
    ra = numpy.empty(shape + (1024,), numpy.int32); rb = numpy.empty(shape, numpy.int32)
    ref.tlwe_cpu.TLweExtractLweSamplesReference(tl, shape)(ra, rb, tl_a)
    out['tlwe_extract_a'] = ra; out['tlwe_extract_b'] = rb
    # mask_size = 2 extract
    tl2 = ref_shim.RefTLweParams(1024, 2)
    tl_a2 = gi.tlwe_extract_inputs(mask_size=2)
    shape = tl_a2.shape[:-2]
    ra = numpy.empty(shape + (2048,), numpy.int32); rb = numpy.empty(shape, numpy.int32)
    ref.tlwe_cpu.TLweExtractLweSamplesReference(tl2, shape)(ra, rb, tl_a2)
    out['tlwe_extract2_a'] = ra; out['tlwe_extract2_b'] = rb

    # ---- TGSW decomposition / MAC / external product (test/test_tgsw.py:44-154) ----
    tg = ref_shim.RefTGswParams(tl, 2, 10)
    sample = gi.decomp_inputs()
    shape = sample.shape[:-2]
    r = numpy.empty(shape + (2, 2, 1024), numpy.int32)
    ref.tgsw_cpu.tgsw_polynomial_decomp_trf_reference(tg, shape)(r, sample)
    out['tgsw_decomp'] = r

    tr_sample, bk, row = gi.mac_inputs()
    shape = tr_sample.shape[:-3]
    r = numpy.empty(shape + (2, 1024), numpy.uint64)
    ref.tgsw_cpu.tlwe_transformed_add_mul_to_trf_reference(tg, shape, bk.shape[0], None)(r, tr_sample, bk, row)
    out['tgsw_mac'] = r
    print("mac done", time.time() - t0)

    accum, bk, row = gi.extmul_inputs()
    acc = accum.copy()
    ref.tgsw_cpu.TGswTransformedExternalMulReference(tg, accum.shape[:-2], bk.shape[0], None)(acc, bk, row)
    out['tgsw_extmul'] = acc
    accum, bk, row = gi.extmul_inputs(full_range=True)
    acc = accum.copy()
    ref.tgsw_cpu.TGswTransformedExternalMulReference(tg, accum.shape[:-2], bk.shape[0], None)(acc, bk, row)
    out['tgsw_extmul_full'] = acc
    print("extmul done", time.time() - t0)

    # ---- keyswitch (test/test_lwe.py:47-101) ----
    ks_a, ks_b, ks_cv, src_a, src_b = gi.keyswitch_inputs()
    shape = src_b.shape
    ra = numpy.empty(shape + (500,), numpy.int32); rb = numpy.empty(shape, numpy.int32)
    rcv = numpy.empty(shape, numpy.float32)
    ref.lwe_cpu.LweKeyswitchReference(None, 1024, 500, 8, 2)(ra, rb, rcv, ks_a, ks_b, ks_cv, src_a, src_b)
    out['ks_a'] = ra; out['ks_b'] = rb; out['ks_cv'] = rcv
    print("keyswitch done", time.time() - t0)

    # ---- LWE linear / trivial (test/test_lwe.py:216-385) ----
    res, src = gi.linear_inputs()
    for p, add in ((1, False), (-1, True), (2, True), (-2, True)):
        ra, rb, rcv = res[0].copy(), res[1].copy(), res[2].copy()
        ref.lwe_cpu.LweLinearReference(None, None, add_result=add)(ra, rb, rcv, src[0], src[1], src[2], p)
        out['linear_p%d_add%d_a' % (p, add)] = ra
        out['linear_p%d_add%d_b' % (p, add)] = rb
        out['linear_p%d_add%d_cv' % (p, add)] = rcv

    # ---- key generation pieces: encrypt_zero, add_message, ks-key, lwe encrypt/decrypt ----
    key, n1, n2 = gi.encrypt_zero_inputs()
    shape = n2.shape[:-1]
    ra = numpy.empty(shape + (2, 1024), numpy.int32); rcv = numpy.empty(shape, numpy.float32)
    ref.tlwe_cpu.TLweEncryptZeroReference(tl, shape, 9e-9, None)(ra, rcv, key, n1, n2)
    out['encrypt_zero_a'] = ra; out['encrypt_zero_cv'] = rcv

    tgsw_a, msgs = gi.add_message_inputs()
    r = tgsw_a.copy()
    ref.tgsw_cpu.TGswAddMessageReference(tg, msgs.shape)(r, msgs)
    out['add_message'] = r

    in_key, out_key, na, nb = gi.ks_keygen_inputs()
    ks_a = numpy.empty((in_key.size, 8, 4, out_key.size), numpy.int32)
    ks_b = numpy.empty((in_key.size, 8, 4), numpy.int32)
    ks_cv = numpy.empty((in_key.size, 8, 4), numpy.float32)
    ref.lwe_cpu.MakeLweKeyswitchKeyReference(in_key.size, out_key.size, 8, 2, 1e-3)(
        ks_a, ks_b, ks_cv, in_key, out_key, na, nb)
    out['kskey_a'] = ks_a; out['kskey_b'] = ks_b; out['kskey_cv'] = ks_cv

    msgs, key, na, nb = gi.lwe_encrypt_inputs()
    ra = numpy.empty(msgs.shape + (500,), numpy.int32); rb = numpy.empty(msgs.shape, numpy.int32)
    rcv = numpy.empty(msgs.shape, numpy.float32)
    ref.lwe_cpu.LweEncryptReference(msgs.shape, 500, 1e-3)(ra, rb, rcv, msgs, key, na, nb)
    out['lwe_encrypt_a'] = ra; out['lwe_encrypt_b'] = rb
    dec = numpy.empty(msgs.shape, numpy.int32)
    ref.lwe_cpu.LweDecryptReference(msgs.shape, 500)(dec, ra, rb, key)
    out['lwe_decrypt'] = dec

    # ---- composition: reduced blind rotate in the reference's driver order ----
    # bootstrap.py:96-142 (mux_rotate loop) + :193 extract, on B=2 bits, n_iter=3
    acc0, bk, bara = gi.blind_rotate_inputs()
    acc = acc0.copy()
    shift = ref.polynomials_cpu.ShiftTorusPolynomialReference(
        1024, acc.shape[:-1], bara.shape, powers_view=True, minus_one=True)
    extmul = ref.tgsw_cpu.TGswTransformedExternalMulReference(tg, acc.shape[:-2], bk.shape[0], None)
    for i in range(bk.shape[0]):
        tmp = numpy.empty_like(acc)
        shift(tmp, acc, bara, i)                  # tlwe_shift_polynomials, bootstrap.py:103
        extmul(tmp, bk, i)                        # bootstrap.py:106
        acc = acc + tmp                           # tlwe_add_to, bootstrap.py:109
    out['blind_rotate_acc'] = acc
    shape = acc.shape[:-2]
    ra = numpy.empty(shape + (1024,), numpy.int32); rb = numpy.empty(shape, numpy.int32)
    ref.tlwe_cpu.TLweExtractLweSamplesReference(tl, shape)(ra, rb, acc)
    out['blind_rotate_ext_a'] = ra; out['blind_rotate_ext_b'] = rb
    print("blind rotate done", time.time() - t0)

    # ---- FFT transform path (BASELINE config 5): the reference's functions with transform_type='FFT'
    ft = ref.fft_transform
    out['fft_forward'] = ft.forward_transform_ref(polys_i32)                 # transform/fft.py:27-51
    out['fft_inverse_of_forward'] = ft.inverse_transform_ref(out['fft_forward'])
    tl_f = ref_shim.RefTLweParams(1024, 1, 'FFT')
    tg_f = ref_shim.RefTGswParams(tl_f, 2, 10)
    accum, tgsw, row = gi.fft_extmul_inputs()
    bkf = ft.forward_transform_ref(tgsw)                                     # TLweTransformSamples (FFT)
    acc = accum.copy()
    ref.tgsw_cpu.TGswTransformedExternalMulReference(tg_f, accum.shape[:-2], bkf.shape[0], None)(acc, bkf, row)
    out['fft_extmul'] = acc

    path = os.path.join(HERE, 'reference_outputs.npz')
    numpy.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; cases:", len(out))


if __name__ == '__main__':
    main()
