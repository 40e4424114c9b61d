"""
golden_inputs.py -- seeded input generators shared by make_golden.py (which feeds them to the
reference's own CPU functions) and by the tests (which feed them to the oracle / the HIP library).
numpy.random.RandomState is bit-stable across NumPy versions, so inputs need not be stored.
Shapes and value ranges mirror the reference's differential tests (test/test_*.py).
"""

import numpy

P = 2**64 - 2**32 + 1


def _rs(seed):
    return numpy.random.RandomState(seed)


def ff_inputs(count=256):
    """test/test_transform/test_arithmetic.py: random field elements + the edge values the
    reference pins (mod at P-1; mul regression (P-1, 2^33) :169-174; lsh regression :243-249)."""
    rs = _rs(1001)
    a = rs.randint(0, P, size=count, dtype=numpy.uint64)
    b = rs.randint(0, P, size=count, dtype=numpy.uint64)
    e = rs.randint(0, 2**32, size=count, dtype=numpy.uint64).astype(numpy.uint32)
    s = rs.randint(0, 192, size=count).astype(numpy.uint32)
    a[:6] = [0, 1, P - 1, P - 1, 11509900421665959066, 2**32]
    b[:6] = [0, P - 1, P - 1, 2**33, 1, 2**32 - 1]
    s[:6] = [0, 31, 32, 95, 191, 96]
    e[:6] = [0, 1, 2, 3, 2**32 - 1, 64]
    return a, b, e, s


def ntt_inputs():
    """test/test_transform/test_computation.py:33-68: full-range i32 polynomials and full-range
    field elements, N=1024."""
    rs = _rs(1002)
    polys_i32 = rs.randint(-2**31, 2**31, size=(3, 1024), dtype=numpy.int32)
    polys_i32[2, :] = 0
    polys_i32[2, 1] = 1          # delta at j=1 -> psi^(2k+1): pins root, ordering and direction
    polys_ff = rs.randint(0, P, size=(2, 1024), dtype=numpy.uint64)
    return polys_i32, polys_ff


def ntt_small_inputs():
    rs = _rs(1003)
    return rs.randint(-2**31, 2**31, size=(2, 16), dtype=numpy.int32)


def modswitch_inputs():
    rs = _rs(1004)
    x = rs.randint(-2**31, 2**31, size=(5, 41), dtype=numpy.int32)
    x.flat[:8] = [0, -1, 2**31 - 1, -2**31, 2**20 - 1, 2**20, -2**20, -2**20 - 1]
    return x


def shift_inputs():
    """test/test_polynomials.py:30-58 (N=16, powers in [0, 2N)) plus an N=1024 case."""
    rs = _rs(1005)
    out = {}
    for tag, N, shape, polys in (('n16', 16, (20,), (3,)), ('n1024', 1024, (7,), (2,))):
        src = rs.randint(-2**31, 2**31, size=shape + polys + (N,), dtype=numpy.int32)
        powers = rs.randint(0, 2 * N, size=shape).astype(numpy.int32)
        powers[:4] = [0, N, N - 1, 2 * N - 1]
        out[tag] = (src, powers, N)
    return out


def shift_view_inputs():
    rs = _rs(1006)
    N = 1024
    src = rs.randint(-2**31, 2**31, size=(3, 2, N), dtype=numpy.int32)
    powers = rs.randint(0, 2 * N, size=(3, 5)).astype(numpy.int32)
    return src, powers, 3, N


def tlwe_trivial_inputs():
    rs = _rs(1007)
    return rs.randint(-2**31, 2**31, size=(2, 3, 1024), dtype=numpy.int32)


def tlwe_extract_inputs(mask_size=1):
    rs = _rs(1008 + mask_size)
    return rs.randint(-2**31, 2**31, size=(2, 3, mask_size + 1, 1024), dtype=numpy.int32)


def decomp_inputs():
    """test/test_tgsw.py:44-69: full-range accumulators, shape (2,3)."""
    rs = _rs(1010)
    x = rs.randint(-2**31, 2**31, size=(2, 3, 2, 1024), dtype=numpy.int32)
    x[0, 0, 0, :8] = [0, -1, 2**31 - 1, -2**31, 2**21, -2**21, 2**11, -2**11 - 1]
    return x


def mac_inputs():
    """test/test_tgsw.py:72-115."""
    rs = _rs(1011)
    tr_sample = rs.randint(0, P, size=(2, 3, 2, 2, 1024), dtype=numpy.uint64)
    bk = rs.randint(0, P, size=(4, 2, 2, 2, 1024), dtype=numpy.uint64)
    return tr_sample, bk, 2


def extmul_inputs(full_range=False):
    """test/test_tgsw.py:118-154: shape (2,3), accum in (-1000,1000) (or full range), random
    transformed-domain BK, row 2."""
    rs = _rs(1012 + int(full_range))
    bk = rs.randint(0, P, size=(4, 2, 2, 2, 1024), dtype=numpy.uint64)
    if full_range:
        accum = rs.randint(-2**31, 2**31, size=(2, 3, 2, 1024), dtype=numpy.int32)
    else:
        accum = rs.randint(-1000, 1000, size=(2, 3, 2, 1024)).astype(numpy.int32)
    return accum, bk, 2


def keyswitch_inputs():
    """test/test_lwe.py:47-101: batch (4,5), KS entries in (-1000,1000), base-0 slice zero."""
    rs = _rs(1014)
    ks_a = rs.randint(-1000, 1000, size=(1024, 8, 4, 500)).astype(numpy.int32)
    ks_b = rs.randint(-1000, 1000, size=(1024, 8, 4)).astype(numpy.int32)
    ks_cv = rs.uniform(-1, 1, size=(1024, 8, 4)).astype(numpy.float32)
    ks_a[:, :, 0, :] = 0
    ks_b[:, :, 0] = 0
    ks_cv[:, :, 0] = 0
    src_a = rs.randint(-2**31, 2**31, size=(4, 5, 1024), dtype=numpy.int32)
    src_b = rs.randint(-1000, 1000, size=(4, 5)).astype(numpy.int32)
    return ks_a, ks_b, ks_cv, src_a, src_b


def linear_inputs():
    """test/test_lwe.py:216-294."""
    rs = _rs(1015)
    shape = (3, 4)
    def lwe():
        return (rs.randint(-2**31, 2**31, size=shape + (500,), dtype=numpy.int32),
                rs.randint(-2**31, 2**31, size=shape, dtype=numpy.int32),
                rs.uniform(0, 1, size=shape).astype(numpy.float32))
    return lwe(), lwe()


def encrypt_zero_inputs():
    """test/test_tlwe.py:97-131."""
    rs = _rs(1016)
    key = rs.randint(0, 2, size=(1, 1024)).astype(numpy.int32)
    n1 = rs.randint(-2**31, 2**31, size=(2, 3, 1, 1024), dtype=numpy.int32)
    n2 = rs.randint(-1000, 1000, size=(2, 3, 1024)).astype(numpy.int32)
    return key, n1, n2


def add_message_inputs():
    """test/test_tgsw.py:157-184."""
    rs = _rs(1017)
    tgsw_a = rs.randint(-2**31, 2**31, size=(5, 2, 2, 2, 1024), dtype=numpy.int32)
    msgs = rs.randint(0, 2, size=(5,)).astype(numpy.int32)
    return tgsw_a, msgs


def ks_keygen_inputs():
    """test/test_lwe.py:104-146 (reduced sizes: input 64, output 50)."""
    rs = _rs(1018)
    in_key = rs.randint(0, 2, size=(64,)).astype(numpy.int32)
    out_key = rs.randint(0, 2, size=(50,)).astype(numpy.int32)
    na = rs.randint(-2**31, 2**31, size=(64, 8, 3, 50), dtype=numpy.int32)
    nb = rs.randint(-2**31, 2**31, size=(64, 8, 3), dtype=numpy.int32)
    return in_key, out_key, na, nb


def lwe_encrypt_inputs():
    """test/test_lwe.py:149-213."""
    rs = _rs(1019)
    shape = (4, 5)
    msgs = rs.randint(-2**31, 2**31, size=shape, dtype=numpy.int32)
    key = rs.randint(0, 2, size=(500,)).astype(numpy.int32)
    na = rs.randint(-2**31, 2**31, size=shape + (500,), dtype=numpy.int32)
    nb = rs.randint(-2**31, 2**31, size=shape, dtype=numpy.int32)
    return msgs, key, na, nb


def blind_rotate_inputs():
    """Reduced blind rotate: B=2 bits, 3 iterations, random transformed-domain BK rows,
    full-range accumulator (worst case for the exactness argument), bara in [0, 2N)."""
    rs = _rs(1020)
    acc = rs.randint(-2**31, 2**31, size=(2, 2, 1024), dtype=numpy.int32)
    bk = rs.randint(0, P, size=(3, 2, 2, 2, 1024), dtype=numpy.uint64)
    bara = rs.randint(0, 2048, size=(2, 3)).astype(numpy.int32)
    bara[0, 0] = 1024 + 17
    bara[1, 1] = 0
    return acc, bk, bara


def fft_extmul_inputs():
    """FFT external product (test/test_tgsw.py:118-154 with transform_type='FFT'), realistic
    magnitudes: coefficient-domain TGSW rows are full-range int32 (transformed by the caller),
    accumulators full-range."""
    rs = _rs(1030)
    tgsw = rs.randint(-2**31, 2**31, size=(3, 2, 2, 2, 1024), dtype=numpy.int32)
    accum = rs.randint(-2**31, 2**31, size=(2, 3, 2, 1024), dtype=numpy.int32)
    return accum, tgsw, 1


def extmul_inputs_k2(full_range=False):
    """extmul_inputs with tlwe_mask_size = 2: BK rows are [3][2][3][1024], accumulators [3][1024]."""
    rs = _rs(1030 + int(full_range))
    bk = rs.randint(0, P, size=(3, 3, 2, 3, 1024), dtype=numpy.uint64)
    if full_range:
        accum = rs.randint(-2**31, 2**31, size=(2, 2, 3, 1024), dtype=numpy.int32)
    else:
        accum = rs.randint(-1000, 1000, size=(2, 2, 3, 1024)).astype(numpy.int32)
    return accum, bk, 1


def encrypt_zero_inputs_k2():
    rs = _rs(1032)
    key = rs.randint(0, 2, size=(2, 1024)).astype(numpy.int32)
    n1 = rs.randint(-2**31, 2**31, size=(2, 3, 2, 1024), dtype=numpy.int32)
    n2 = rs.randint(-1000, 1000, size=(2, 3, 1024)).astype(numpy.int32)
    return key, n1, n2
