"""
oracle_fft.py -- CPU oracle of the FFT transform path (BASELINE config 5).  TEST INFRASTRUCTURE.

NumPy restatement of the reference's fp64 negacyclic "tangent" FFT and of the path composed on
top of it; numpy.fft is the reference's own CPU engine for this path (nufhe/transform/fft.py:42,51).
The integer pieces (decomposition, rotation, extract, keyswitch, mod-switch) are shared with the
C oracle (oracle/oracle.py).

fp64 results are not bit-reproducible across FFT implementations (the reference asserts `allclose`
for transforms, test/test_transform/test_computation.py:67-68); parity for this path is therefore
stated as a tolerance and checked against the exact NTT oracle.
"""

import numpy

from . import oracle as orc

N = 1024
_COEFFS = numpy.exp(-2j * numpy.pi * numpy.arange(N // 2) / N / 2)     # transform/fft.py:37


def fft_forward(data):
    """forward_transform_ref: int32 [..., 1024] -> complex128 [..., 512]
    (transform/fft.py:48-51, polynomial_transform_fft.py:55-56)"""
    data = numpy.asarray(data)
    z = (data[..., :N // 2] - 1j * data[..., N // 2:]) * _COEFFS
    return numpy.fft.fft(z, axis=-1)


def fft_inverse(data):
    """inverse_transform_ref: complex128 [..., 512] -> int32 [..., 1024]
    (transform/fft.py:39-47, polynomial_transform_fft.py:59-60)"""
    res = numpy.fft.ifft(numpy.asarray(data), axis=-1).conj() * _COEFFS
    f64_to_i32 = lambda x: numpy.round(x).astype(numpy.int64).astype(numpy.int32)
    return numpy.concatenate([f64_to_i32(res.real), f64_to_i32(res.imag)], axis=-1)


def bk_from_coeffs(tgsw):
    """TLweTransformSamples for FFT: forward transform only (tlwe_gpu.py:199-236; prepare = identity,
    polynomial_transform_fft.py:95-105).  tgsw int32 [n, 2, 2, 2, 1024] -> complex128 [n, 2, 2, 2, 512]"""
    return fft_forward(tgsw)


def external_mul(accum, bk, row):
    """TGswTransformedExternalMulReference with the FFT transform (tgsw_cpu.py:82-106).
    accum int32 [..., 2, 1024]; bk complex128 [n, 2, 2, 2, 512]"""
    dec = orc.tgsw_decomp(accum)                       # [..., 2, 2, 1024]
    tr = fft_forward(dec)                              # [..., 2, 2, 512]
    out = numpy.zeros(accum.shape[:-1] + (N // 2,), numpy.complex128)
    for mi in range(2):
        for d in range(2):
            out = out + tr[..., mi, d, None, :] * bk[row, mi, d, :, :]     # tgsw_cpu.py:69-77
    return fft_inverse(out)


def blind_rotate(accum, bk, bara, n_iter=None):
    """bootstrap.py:119-142 with the FFT external product."""
    acc = numpy.ascontiguousarray(accum, numpy.int32).copy()
    n_iter = bara.shape[-1] if n_iter is None else n_iter
    for i in range(n_iter):
        tmp = orc.shift_torus_polynomial(acc, bara[..., i], minus_one=True)
        tmp = external_mul(tmp, bk, i)
        acc = (acc + tmp).astype(numpy.int32)
    return acc


def bootstrap_extract(bk, xa, xb, mu):
    """bootstrap(..., no_keyswitch=True), bootstrap.py:206-229,154-196."""
    xa = numpy.asarray(xa, numpy.int32); xb = numpy.asarray(xb, numpy.int32)
    barb = orc.t32_to_phase(xb, 2 * N)
    bara = orc.t32_to_phase(xa, 2 * N)
    tv = numpy.full(xb.shape + (N,), mu, numpy.int32)
    tvb = orc.shift_torus_polynomial(tv, barb, invert_powers=True)
    acc, _ = orc.tlwe_noiseless_trivial(tvb, 1)
    acc = blind_rotate(acc, bk, bara)
    return orc.tlwe_extract_lwe_samples(acc)


MU = 2**29


def gate_binary(bk, ck, a, b, c, pa, pb, mu=MU):
    """Binary gate with the FFT bootstrap; ck supplies the keyswitch key arrays (oracle.CloudKeyArrays)."""
    ta = (numpy.int32(pa) * a[0] + numpy.int32(pb) * b[0]).astype(numpy.int32)
    tb = (numpy.int32(c) + numpy.int32(pa) * a[1] + numpy.int32(pb) * b[1]).astype(numpy.int32)
    ea, eb = bootstrap_extract(bk, ta, tb, mu)
    return orc.lwe_keyswitch(ck.ks_a, ck.ks_b, ck.ks_cv, ea, eb)


def gate(name, bk, ck, a, b):
    c, pa, pb = orc.BINARY_GATES[name]
    return gate_binary(bk, ck, a, b, orc._wrap32(c), pa, pb)


def gate_mux(bk, ck, a, b, c):
    """gates.py:600-664"""
    m = numpy.int32(MU)
    ta = (a[0] + b[0]).astype(numpy.int32); tb = (-m + a[1] + b[1]).astype(numpy.int32)
    u1a, u1b = bootstrap_extract(bk, ta, tb, MU)
    ta = (-a[0] + c[0]).astype(numpy.int32); tb = (-m - a[1] + c[1]).astype(numpy.int32)
    u2a, u2b = bootstrap_extract(bk, ta, tb, MU)
    sa = (u1a + u2a).astype(numpy.int32); sb = (m + u1b + u2b).astype(numpy.int32)
    return orc.lwe_keyswitch(ck.ks_a, ck.ks_b, ck.ks_cv, sa, sb)


def tgsw_coeffs_from_reference_bk(bk_ntt):
    """Recovers the coefficient-domain TGSW samples from the NTT-format key of oracle.make_key_pair
    (inverse NTT of the un-Montgomery'd key), so that the FFT key encrypts the same secret with the
    same noise: key generation itself is transform-independent because the products of
    TLweEncryptZero are exact in both transforms (|values| <= 2^41 << 2^52)."""
    plain = orc.ff_mul(bk_ntt, numpy.uint64(0xfffffffe00000001))     # x * 2^-64
    return orc.ntt_inverse(plain, i32_conversion=True)
