"""
oracle.py -- ctypes front-end of the C CPU oracle (oracle/nufhe_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package ``nufhe_amd`` must never import this module.

The functions take and return NumPy arrays and follow the argument meaning of the reference's
``*Reference`` factories (nufhe/*_cpu.py); the whole-gate helpers compose them in the order of
the reference's multi-kernel driver (nufhe/bootstrap.py:96-229, nufhe/gates.py:81-121,600-664).
The RNG-order-faithful key generation (`make_key_pair`, `encrypt`) follows SURVEY App. D.
"""

import ctypes
import os
import subprocess

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnufhe_oracle.so")

P = 2**64 - 2**32 + 1

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "nufhe_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libnufhe_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_ff_root_of_unity.restype = ctypes.c_uint64
        _lib.orc_ff_root_of_unity.argtypes = [ctypes.c_uint64]
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def _p(arr, typ):
    return arr.ctypes.data_as(typ)


def _i32(x):
    return numpy.ascontiguousarray(x, dtype=numpy.int32)


def _u64(x):
    return numpy.ascontiguousarray(x, dtype=numpy.uint64)


def _u32(x):
    return numpy.ascontiguousarray(x, dtype=numpy.uint32)


def _f32(x):
    return numpy.ascontiguousarray(x, dtype=numpy.float32)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(ctypes.c_int(n))


# ---------------------------------------------------------------- finite field

def _ff_binary(name, a, b):
    a, b = numpy.broadcast_arrays(_u64(a), _u64(b))
    a, b = _u64(a), _u64(b)
    r = numpy.empty(a.shape, numpy.uint64)
    getattr(lib(), name)(_p(r, c_u64p), _p(a, c_u64p), _p(b, c_u64p), ctypes.c_long(a.size))
    return r


def ff_add(a, b): return _ff_binary("orc_ff_add", a, b)
def ff_sub(a, b): return _ff_binary("orc_ff_sub", a, b)
def ff_mul(a, b): return _ff_binary("orc_ff_mul", a, b)
def ff_mul_prepared(a, b): return _ff_binary("orc_ff_mul_prepared", a, b)


def _ff_unary(name, a):
    a = _u64(a)
    r = numpy.empty(a.shape, numpy.uint64)
    getattr(lib(), name)(_p(r, c_u64p), _p(a, c_u64p), ctypes.c_long(a.size))
    return r


def ff_mod(a): return _ff_unary("orc_ff_mod", a)
def ff_prepare_for_mul(a): return _ff_unary("orc_ff_prepare_for_mul", a)


def ff_pow(a, e):
    a = _u64(a); e = _u32(numpy.broadcast_to(e, a.shape))
    r = numpy.empty(a.shape, numpy.uint64)
    lib().orc_ff_pow(_p(r, c_u64p), _p(a, c_u64p), _p(e, c_u32p), ctypes.c_long(a.size))
    return r


def ff_inv_pow2(e):
    e = _u32(e)
    r = numpy.empty(e.shape, numpy.uint64)
    lib().orc_ff_inv_pow2(_p(r, c_u64p), _p(e, c_u32p), ctypes.c_long(e.size))
    return r


def ff_lsh(a, s):
    a = _u64(a); s = _u32(numpy.broadcast_to(s, a.shape))
    r = numpy.empty(a.shape, numpy.uint64)
    lib().orc_ff_lsh(_p(r, c_u64p), _p(a, c_u64p), _p(s, c_u32p), ctypes.c_long(a.size))
    return r


def ff_to_i32(a):
    a = _u64(a)
    r = numpy.empty(a.shape, numpy.int32)
    lib().orc_ff_to_i32(_p(r, c_i32p), _p(a, c_u64p), ctypes.c_long(a.size))
    return r


def ff_from_i32(a):
    a = _i32(a)
    r = numpy.empty(a.shape, numpy.uint64)
    lib().orc_ff_from_i32(_p(r, c_u64p), _p(a, c_i32p), ctypes.c_long(a.size))
    return r


def root_of_unity(order):
    return int(lib().orc_ff_root_of_unity(ctypes.c_uint64(order)))


# ---------------------------------------------------------------- transforms

def ntt_forward(data, i32_conversion=True):
    """ntt_transform_ref(data, i32_conversion=...) -- nufhe/transform/ntt.py:30-44."""
    n = data.shape[-1]
    batch = data.size // n
    out = numpy.empty(data.shape, numpy.uint64)
    if i32_conversion:
        d = _i32(data)
        lib().orc_ntt_forward_i32(_p(out, c_u64p), _p(d, c_i32p), ctypes.c_long(batch), ctypes.c_int(n))
    else:
        d = _u64(data)
        lib().orc_ntt_forward_u64(_p(out, c_u64p), _p(d, c_u64p), ctypes.c_long(batch), ctypes.c_int(n))
    return out


def ntt_inverse(data, i32_conversion=True):
    n = data.shape[-1]
    batch = data.size // n
    d = _u64(data)
    if i32_conversion:
        out = numpy.empty(data.shape, numpy.int32)
        lib().orc_ntt_inverse_i32(_p(out, c_i32p), _p(d, c_u64p), ctypes.c_long(batch), ctypes.c_int(n))
    else:
        out = numpy.empty(data.shape, numpy.uint64)
        lib().orc_ntt_inverse_u64(_p(out, c_u64p), _p(d, c_u64p), ctypes.c_long(batch), ctypes.c_int(n))
    return out


def poly_mul_schoolbook(a, b):
    """Negacyclic product mod 2^32 (test/test_transform/test_computation.py poly_mul_ref)."""
    a = _i32(a); b = _i32(b)
    n = a.shape[-1]
    out = numpy.empty(a.shape, numpy.int32)
    lib().orc_poly_mul_schoolbook(_p(out, c_i32p), _p(a, c_i32p), _p(b, c_i32p),
                                  ctypes.c_long(a.size // n), ctypes.c_int(n))
    return out


# ---------------------------------------------------------------- small ops

def t32_to_phase(phase, mspace_size):
    phase = _i32(phase)
    out = numpy.empty(phase.shape, numpy.int32)
    lib().orc_t32_to_phase(_p(out, c_i32p), _p(phase, c_i32p), ctypes.c_long(phase.size),
                           ctypes.c_uint32(mspace_size))
    return out


def shift_torus_polynomial(source, powers, minus_one=False, invert_powers=False):
    """source [batch..., polys..., N], powers [batch...]; polynomials_cpu.py:25-59."""
    source = _i32(source); powers = _i32(powers)
    N = source.shape[-1]
    batch = powers.size
    polys = source.size // (batch * N)
    out = numpy.empty(source.shape, numpy.int32)
    lib().orc_shift_torus_polynomial(
        _p(out, c_i32p), _p(source, c_i32p), _p(powers, c_i32p), ctypes.c_long(batch),
        ctypes.c_int(polys), ctypes.c_int(N), ctypes.c_int(int(minus_one)), ctypes.c_int(int(invert_powers)))
    return out


def tlwe_noiseless_trivial(mu, mask_size):
    mu = _i32(mu)
    N = mu.shape[-1]
    batch = mu.size // N
    a = numpy.empty(mu.shape[:-1] + (mask_size + 1, N), numpy.int32)
    cv = numpy.empty(mu.shape[:-1], numpy.float32)
    lib().orc_tlwe_noiseless_trivial(_p(a, c_i32p), _p(cv, c_f32p), _p(mu, c_i32p),
                                     ctypes.c_long(batch), ctypes.c_int(mask_size), ctypes.c_int(N))
    return a, cv


def tlwe_extract_lwe_samples(tlwe_a):
    tlwe_a = _i32(tlwe_a)
    N = tlwe_a.shape[-1]; k1 = tlwe_a.shape[-2]; k = k1 - 1
    shape = tlwe_a.shape[:-2]
    batch = tlwe_a.size // (k1 * N)
    ra = numpy.empty(shape + (k * N,), numpy.int32)
    rb = numpy.empty(shape, numpy.int32)
    lib().orc_tlwe_extract_lwe_samples(_p(ra, c_i32p), _p(rb, c_i32p), _p(tlwe_a, c_i32p),
                                       ctypes.c_long(batch), ctypes.c_int(k), ctypes.c_int(N))
    return ra, rb


def tgsw_decomp(sample, decomp_length=2, log2_base=10):
    """sample [..., k+1, N] -> [..., k+1, l, N]; tgsw_cpu.py:26-49."""
    sample = _i32(sample)
    N = sample.shape[-1]; k1 = sample.shape[-2]
    batch = sample.size // (k1 * N)
    out = numpy.empty(sample.shape[:-1] + (decomp_length, N), numpy.int32)
    lib().orc_tgsw_decomp(_p(out, c_i32p), _p(sample, c_i32p), ctypes.c_long(batch),
                          ctypes.c_int(k1 - 1), ctypes.c_int(decomp_length), ctypes.c_int(log2_base),
                          ctypes.c_int(N))
    return out


def tlwe_transformed_add_mul(sample, bk, bk_row):
    """sample u64 [..., k+1, l, N]; bk u64 [n, k+1, l, k+1, N]; tgsw_cpu.py:52-79."""
    sample = _u64(sample); bk = _u64(bk)
    N = sample.shape[-1]; l = sample.shape[-2]; k1 = sample.shape[-3]
    batch = sample.size // (k1 * l * N)
    out = numpy.empty(sample.shape[:-3] + (k1, N), numpy.uint64)
    lib().orc_tlwe_transformed_add_mul(_p(out, c_u64p), _p(sample, c_u64p), _p(bk, c_u64p),
                                       ctypes.c_int(bk_row), ctypes.c_long(batch), ctypes.c_int(k1 - 1),
                                       ctypes.c_int(l), ctypes.c_int(N))
    return out


def tgsw_external_mul(accum, bk, bk_row, log2_base=10):
    """accum i32 [..., k+1, N] (returned, not in-place); tgsw_cpu.py:82-106."""
    accum = _i32(accum).copy(); bk = _u64(bk)
    N = accum.shape[-1]; k1 = accum.shape[-2]; l = bk.shape[2]
    batch = accum.size // (k1 * N)
    lib().orc_tgsw_external_mul(_p(accum, c_i32p), _p(bk, c_u64p), ctypes.c_int(bk_row),
                                ctypes.c_long(batch), ctypes.c_int(k1 - 1), ctypes.c_int(l),
                                ctypes.c_int(log2_base), ctypes.c_int(N))
    return accum


def blind_rotate(accum, bk, bara, n_iter=None, log2_base=10):
    """bootstrap.py:119-142; accum i32 [batch, k+1, N], bara i32 [batch, n]."""
    accum = _i32(accum).copy(); bk = _u64(bk); bara = _i32(bara)
    N = accum.shape[-1]; k1 = accum.shape[-2]; l = bk.shape[2]
    batch = accum.size // (k1 * N)
    if n_iter is None:
        n_iter = bara.shape[-1]
    lib().orc_blind_rotate(_p(accum, c_i32p), _p(bk, c_u64p), _p(bara, c_i32p),
                           ctypes.c_long(bara.shape[-1]), ctypes.c_int(n_iter), ctypes.c_long(batch),
                           ctypes.c_int(k1 - 1), ctypes.c_int(l), ctypes.c_int(log2_base), ctypes.c_int(N))
    return accum


def lwe_keyswitch(ks_a, ks_b, ks_cv, source_a, source_b):
    """lwe_cpu.py:62-93; ks_a [in, t, base, out]."""
    ks_a = _i32(ks_a); ks_b = _i32(ks_b); ks_cv = _f32(ks_cv)
    source_a = _i32(source_a); source_b = _i32(source_b)
    input_size, t, base, output_size = ks_a.shape
    log2_base = int(numpy.log2(base))
    batch = source_b.size
    ra = numpy.empty(source_b.shape + (output_size,), numpy.int32)
    rb = numpy.empty(source_b.shape, numpy.int32)
    rcv = numpy.empty(source_b.shape, numpy.float32)
    lib().orc_lwe_keyswitch(_p(ra, c_i32p), _p(rb, c_i32p), _p(rcv, c_f32p),
                            _p(ks_a, c_i32p), _p(ks_b, c_i32p), _p(ks_cv, c_f32p),
                            _p(source_a, c_i32p), _p(source_b, c_i32p),
                            ctypes.c_long(batch), ctypes.c_int(input_size), ctypes.c_int(output_size),
                            ctypes.c_int(t), ctypes.c_int(log2_base))
    return ra, rb, rcv


def lwe_linear(res, src, p, add_result=False):
    """res/src = (a, b, cv) tuples of equal shapes; lwe_cpu.py:115-123. Returns new (a, b, cv)."""
    ra, rb, rcv = _i32(res[0]).copy(), _i32(res[1]).copy(), _f32(res[2]).copy()
    sa, sb, scv = (numpy.broadcast_to(_i32(src[0]), ra.shape), numpy.broadcast_to(_i32(src[1]), rb.shape),
                   numpy.broadcast_to(_f32(src[2]), rcv.shape))
    sa, sb, scv = _i32(sa), _i32(sb), _f32(scv)
    n = ra.shape[-1]
    lib().orc_lwe_linear(_p(ra, c_i32p), _p(rb, c_i32p), _p(rcv, c_f32p),
                         _p(sa, c_i32p), _p(sb, c_i32p), _p(scv, c_f32p),
                         ctypes.c_int32(p), ctypes.c_int(int(add_result)), ctypes.c_long(rb.size),
                         ctypes.c_int(n))
    return ra, rb, rcv


def bootstrap_extract(bk, xa, xb, mu, log2_base=10):
    """bootstrap(..., no_keyswitch=True): bootstrap.py:206-229,154-196."""
    bk = _u64(bk); xa = _i32(xa); xb = _i32(xb)
    n = xa.shape[-1]
    batch = xb.size
    _, k1, l, _, N = bk.shape
    ea = numpy.empty(xb.shape + ((k1 - 1) * N,), numpy.int32)
    eb = numpy.empty(xb.shape, numpy.int32)
    lib().orc_bootstrap_extract(_p(ea, c_i32p), _p(eb, c_i32p), _p(bk, c_u64p), _p(xa, c_i32p),
                                _p(xb, c_i32p), ctypes.c_int32(mu), ctypes.c_long(batch), ctypes.c_int(n),
                                ctypes.c_int(k1 - 1), ctypes.c_int(l), ctypes.c_int(log2_base), ctypes.c_int(N))
    return ea, eb


# ---------------------------------------------------------------- whole gates

def _wrap32(x):
    return int(numpy.array(x & 0xffffffff, dtype=numpy.uint32).astype(numpy.int32))


MU = 2**29

# (c, pa, pb) of res = (0, c) + pa * a + pb * b, then bootstrap(MU): gates.py:81-597
BINARY_GATES = {
    'gate_nand': (MU, -1, -1),
    'gate_or': (MU, 1, 1),
    'gate_and': (-MU, 1, 1),
    'gate_nor': (-MU, -1, -1),
    'gate_xor': (2 * MU, 2, 2),
    'gate_xnor': (-2 * MU, -2, -2),
    'gate_andny': (-MU, -1, 1),
    'gate_andyn': (-MU, 1, -1),
    'gate_orny': (MU, -1, 1),
    'gate_oryn': (MU, 1, -1),
}


class CloudKeyArrays:
    """Host arrays of a cloud key in the REFERENCE's formats:
    bk: u64 [n, k+1, l, k+1, N] natural-order NTT, Montgomery-prepared (bootstrap.py:72-76);
    ks_a i32 [kN, t, base, n], ks_b i32 [kN, t, base], ks_cv f32 [kN, t, base] (lwe.py:254-295)."""

    def __init__(self, bk, ks_a, ks_b, ks_cv, bs_log2_base=10):
        self.bk = bk
        self.ks_a = ks_a
        self.ks_b = ks_b
        self.ks_cv = ks_cv
        self.bs_log2_base = bs_log2_base


def gate_binary(ck, a, b, c, pa, pb, mu=MU):
    """a, b = (a_arr [B, n], b_arr [B]); returns (a, b, cv). gates.py:81-121 and siblings."""
    a_a, a_b = _i32(a[0]), _i32(a[1]); b_a, b_b = _i32(b[0]), _i32(b[1])
    n = a_a.shape[-1]; batch = a_b.size
    _, k1, l, _, N = ck.bk.shape
    kN, t, base, _ = ck.ks_a.shape
    ra = numpy.empty(a_a.shape, numpy.int32); rb = numpy.empty(a_b.shape, numpy.int32)
    rcv = numpy.empty(a_b.shape, numpy.float32)
    lib().orc_gate_binary(
        _p(ra, c_i32p), _p(rb, c_i32p), _p(rcv, c_f32p),
        _p(ck.bk, c_u64p), _p(ck.ks_a, c_i32p), _p(ck.ks_b, c_i32p), _p(ck.ks_cv, c_f32p),
        _p(a_a, c_i32p), _p(a_b, c_i32p), _p(b_a, c_i32p), _p(b_b, c_i32p),
        ctypes.c_int32(_wrap32(c)), ctypes.c_int32(pa), ctypes.c_int32(pb), ctypes.c_int32(_wrap32(mu)),
        ctypes.c_long(batch), ctypes.c_int(n), ctypes.c_int(k1 - 1), ctypes.c_int(l),
        ctypes.c_int(ck.bs_log2_base), ctypes.c_int(N), ctypes.c_int(t), ctypes.c_int(int(numpy.log2(base))))
    return ra, rb, rcv


def gate(name, ck, a, b):
    c, pa, pb = BINARY_GATES[name]
    return gate_binary(ck, a, b, c, pa, pb)


def gate_mux(ck, a, b, c):
    arrs = [(_i32(x[0]), _i32(x[1])) for x in (a, b, c)]
    n = arrs[0][0].shape[-1]; batch = arrs[0][1].size
    _, k1, l, _, N = ck.bk.shape
    kN, t, base, _ = ck.ks_a.shape
    ra = numpy.empty(arrs[0][0].shape, numpy.int32); rb = numpy.empty(arrs[0][1].shape, numpy.int32)
    rcv = numpy.empty(arrs[0][1].shape, numpy.float32)
    lib().orc_gate_mux(
        _p(ra, c_i32p), _p(rb, c_i32p), _p(rcv, c_f32p),
        _p(ck.bk, c_u64p), _p(ck.ks_a, c_i32p), _p(ck.ks_b, c_i32p), _p(ck.ks_cv, c_f32p),
        _p(arrs[0][0], c_i32p), _p(arrs[0][1], c_i32p), _p(arrs[1][0], c_i32p), _p(arrs[1][1], c_i32p),
        _p(arrs[2][0], c_i32p), _p(arrs[2][1], c_i32p),
        ctypes.c_long(batch), ctypes.c_int(n), ctypes.c_int(k1 - 1), ctypes.c_int(l),
        ctypes.c_int(ck.bs_log2_base), ctypes.c_int(N), ctypes.c_int(t), ctypes.c_int(int(numpy.log2(base))))
    return ra, rb, rcv


# ---------------------------------------------------------------- key generation / client side

def double_to_t32(d):
    """numeric_functions.py:39-40 (values here are tiny Gaussians; no wrap issue)."""
    d = numpy.asarray(d, numpy.float64)
    return ((d - numpy.trunc(d)) * 2**32).astype(numpy.int32)


class DeterministicRNG:
    """random_numbers.py:46-62."""

    def __init__(self, seed=None):
        self.rng = numpy.random.RandomState(seed)

    def uniform_bool(self, shape):
        return self.rng.randint(0, 2, size=shape, dtype=numpy.int32)

    def uniform_torus32(self, shape):
        return self.rng.randint(-2**31, 2**31, size=shape, dtype=numpy.int32)

    def gauss(self, shape, std_dev):
        return self.rng.normal(size=shape, scale=std_dev)


def rand_gaussian_torus32(rng, message, sigma, shape, centered=False):
    rfloats = rng.gauss(shape, sigma)                      # random_numbers.py:134-139
    if centered:
        rfloats -= rfloats.mean()
    return (numpy.int32(message) + double_to_t32(rfloats)).astype(numpy.int32)


def tlwe_encrypt_zero(key, noises1, noises2, noise):
    """tlwe_cpu.py:64-89; key i32 [k, N]; noises1 [..., k, N]; noises2 [..., N]."""
    key = _i32(key); noises1 = _i32(noises1); noises2 = _i32(noises2)
    k, N = key.shape
    shape = noises2.shape[:-1]
    batch = noises2.size // N
    ra = numpy.empty(shape + (k + 1, N), numpy.int32)
    rcv = numpy.empty(shape, numpy.float32)
    lib().orc_tlwe_encrypt_zero(_p(ra, c_i32p), _p(rcv, c_f32p), _p(key, c_i32p), _p(noises1, c_i32p),
                                _p(noises2, c_i32p), ctypes.c_double(noise), ctypes.c_long(batch),
                                ctypes.c_int(k), ctypes.c_int(N))
    return ra, rcv


def tgsw_add_message(result_a, messages, log2_base=10):
    """result_a i32 [n, k+1, l, k+1, N] modified copy; tgsw_cpu.py:109-126."""
    result_a = _i32(result_a).copy(); messages = _i32(messages)
    n, k1, l, _, N = result_a.shape
    lib().orc_tgsw_add_message(_p(result_a, c_i32p), _p(messages, c_i32p), ctypes.c_long(n),
                               ctypes.c_int(k1 - 1), ctypes.c_int(l), ctypes.c_int(log2_base), ctypes.c_int(N))
    return result_a


def tlwe_transform_samples(values):
    """forward NTT + Montgomery prepare; tlwe_gpu.py:199-236."""
    values = _i32(values)
    N = values.shape[-1]
    out = numpy.empty(values.shape, numpy.uint64)
    lib().orc_tlwe_transform_samples(_p(out, c_u64p), _p(values, c_i32p), ctypes.c_long(values.size // N),
                                     ctypes.c_int(N))
    return out


def make_lwe_keyswitch_key(in_key, out_key, noises_a, noises_b, noise, t, log2_base):
    """lwe_cpu.py:27-59."""
    in_key = _i32(in_key); out_key = _i32(out_key); noises_a = _i32(noises_a); noises_b = _i32(noises_b)
    input_size = in_key.size; output_size = out_key.size; base = 2**log2_base
    ks_a = numpy.empty((input_size, t, base, output_size), numpy.int32)
    ks_b = numpy.empty((input_size, t, base), numpy.int32)
    ks_cv = numpy.empty((input_size, t, base), numpy.float32)
    lib().orc_make_lwe_keyswitch_key(
        _p(ks_a, c_i32p), _p(ks_b, c_i32p), _p(ks_cv, c_f32p), _p(in_key, c_i32p), _p(out_key, c_i32p),
        _p(noises_a, c_i32p), _p(noises_b, c_i32p), ctypes.c_double(noise), ctypes.c_int(input_size),
        ctypes.c_int(output_size), ctypes.c_int(t), ctypes.c_int(log2_base))
    return ks_a, ks_b, ks_cv


def lwe_encrypt(messages, key, noises_a, noises_b, noise):
    """lwe_cpu.py:96-104."""
    messages = _i32(messages); key = _i32(key); noises_a = _i32(noises_a); noises_b = _i32(noises_b)
    n = key.size
    ra = numpy.empty(messages.shape + (n,), numpy.int32)
    rb = numpy.empty(messages.shape, numpy.int32)
    rcv = numpy.empty(messages.shape, numpy.float32)
    lib().orc_lwe_encrypt(_p(ra, c_i32p), _p(rb, c_i32p), _p(rcv, c_f32p), _p(messages, c_i32p),
                          _p(key, c_i32p), _p(noises_a, c_i32p), _p(noises_b, c_i32p),
                          ctypes.c_double(noise), ctypes.c_long(messages.size), ctypes.c_int(n))
    return ra, rb, rcv


def lwe_decrypt(la, lb, key):
    """lwe_cpu.py:107-112."""
    la = _i32(la); lb = _i32(lb); key = _i32(key)
    out = numpy.empty(lb.shape, numpy.int32)
    lib().orc_lwe_decrypt(_p(out, c_i32p), _p(la, c_i32p), _p(lb, c_i32p), _p(key, c_i32p),
                          ctypes.c_long(lb.size), ctypes.c_int(key.size))
    return out


class Params:
    """Scheme constants, api_low_level.py:44-61."""

    def __init__(self, lwe_size=500, N=1024, mask_size=1, bs_decomp_length=2, bs_log2_base=10,
                 ks_decomp_length=8, ks_log2_base=2):
        coeff = (2 / numpy.pi)**0.5
        self.n = lwe_size
        self.N = N
        self.k = mask_size
        self.l = bs_decomp_length
        self.bs_log2_base = bs_log2_base
        self.ks_t = ks_decomp_length
        self.ks_log2_base = ks_log2_base
        self.ks_stdev = 1 / 2**15 * coeff
        self.bs_stdev = 9e-9 * coeff
        self.max_stdev = 1 / 2**4 / 4 * coeff


def make_key_pair(rng, params=None):
    """Host key generation in the reference's RNG consumption order:
    api_low_level.py:174-196, bootstrap.py:59-76, tlwe.py:185-196, lwe.py:265-295.
    Returns (lwe_key i32 [n], tlwe_key i32 [k, N], CloudKeyArrays)."""
    p = params or Params()
    lwe_key = rng.uniform_bool((p.n,))                                        # lwe.py:79
    tlwe_key = rng.uniform_bool((p.k, p.N))                                   # tlwe.py:89
    shape = (p.n, p.k + 1, p.l)
    noises1 = rng.uniform_torus32(shape + (p.k, p.N))                         # tlwe.py:192
    noises2 = rand_gaussian_torus32(rng, 0, p.bs_stdev, shape + (p.N,))       # tlwe.py:193
    tgsw, _ = tlwe_encrypt_zero(tlwe_key, noises1, noises2, p.bs_stdev)       # tgsw.py:146-148
    tgsw = tgsw_add_message(tgsw, lwe_key, p.bs_log2_base)                    # tgsw.py:160-161
    bk = tlwe_transform_samples(tgsw)                                         # bootstrap.py:73-74
    extracted_key = tlwe_key.ravel()                                          # lwe.py:83-90
    base = 2**p.ks_log2_base
    noises_b = rand_gaussian_torus32(
        rng, 0, p.ks_stdev, (p.k * p.N, p.ks_t, base - 1), centered=True)     # lwe.py:285-286
    noises_a = rng.uniform_torus32((p.k * p.N, p.ks_t, base - 1, p.n))        # lwe.py:287-288
    ks_a, ks_b, ks_cv = make_lwe_keyswitch_key(
        extracted_key, lwe_key, noises_a, noises_b, p.ks_stdev, p.ks_t, p.ks_log2_base)
    return lwe_key, tlwe_key, CloudKeyArrays(bk, ks_a, ks_b, ks_cv, p.bs_log2_base)


def encrypt(rng, lwe_key, message, params=None):
    """api_low_level.py:266-281, lwe.py:325-333. Returns (a, b, cv)."""
    p = params or Params()
    message = numpy.asarray(message).astype(bool)
    mus = numpy.where(message, numpy.int32(MU), numpy.int32(-MU)).astype(numpy.int32)
    noises_b = rand_gaussian_torus32(rng, 0, p.ks_stdev, message.shape)
    noises_a = rng.uniform_torus32(message.shape + (p.n,))
    return lwe_encrypt(mus, lwe_key, noises_a, noises_b, p.ks_stdev)


def decrypt(lwe_key, ct):
    """api_low_level.py:261-263,284-295."""
    return lwe_decrypt(ct[0], ct[1], lwe_key) > 0
