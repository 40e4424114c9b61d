"""ctypes loader of the CPU emulation of the wave-level device code (tests only)."""
import ctypes
import os
import subprocess

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_i32p = ctypes.POINTER(ctypes.c_int32)


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", _HERE, "-s", "libnufhe_emu.so"])
        _lib = ctypes.CDLL(os.path.join(_HERE, "libnufhe_emu.so"))
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def ff_binary(name, a, b):
    a = numpy.ascontiguousarray(a, numpy.uint64); b = numpy.ascontiguousarray(b, numpy.uint64)
    r = numpy.empty_like(a)
    getattr(lib(), name)(_p(r, c_u64p), _p(a, c_u64p), _p(b, c_u64p), ctypes.c_long(a.size))
    return r


def ff_lsh_const(a, s):
    a = numpy.ascontiguousarray(a, numpy.uint64)
    r = numpy.empty_like(a)
    lib().emu_ff_lsh_const(_p(r, c_u64p), _p(a, c_u64p), ctypes.c_int(s), ctypes.c_long(a.size))
    return r


def ff_lsh_var(a, s):
    a = numpy.ascontiguousarray(a, numpy.uint64); s = numpy.ascontiguousarray(s, numpy.uint32)
    r = numpy.empty_like(a)
    lib().emu_ff_lsh_var(_p(r, c_u64p), _p(a, c_u64p), _p(s, c_u32p), ctypes.c_long(a.size))
    return r


def ntt_forward(x):
    x = numpy.ascontiguousarray(x, numpy.uint64)
    r = numpy.empty_like(x)
    lib().emu_ntt_forward(_p(r, c_u64p), _p(x, c_u64p))
    return r


def ntt_inverse(x):
    x = numpy.ascontiguousarray(x, numpy.uint64)
    r = numpy.empty_like(x)
    lib().emu_ntt_inverse(_p(r, c_u64p), _p(x, c_u64p))
    return r


def l4_op(op, a, b=None, c=None, shift=0):
    """ff24.h primitives via the shared dispatcher (csrc/l4_hook.h); arrays uint32 [n, 4].
    Returns (out, out2)."""
    a = numpy.ascontiguousarray(a, numpy.uint32)
    n = a.shape[0]
    out = numpy.zeros((n, 4), numpy.uint32); out2 = numpy.zeros((n, 4), numpy.uint32)
    b = None if b is None else numpy.ascontiguousarray(b, numpy.uint32)
    c = None if c is None else numpy.ascontiguousarray(c, numpy.uint32)
    lib().emu_l4_op(_p(out, c_u32p), _p(out2, c_u32p), _p(a, c_u32p), None if b is None else _p(b, c_u32p),
                    None if c is None else _p(c, c_u32p), ctypes.c_int(op), ctypes.c_int(shift), ctypes.c_long(n))
    return out, out2


def ntt_forward_small_l4(d):
    d = numpy.ascontiguousarray(d, numpy.int32)
    r = numpy.empty(1024, numpy.uint64)
    lib().emu_ntt_forward_small_l4(_p(r, c_u64p), _p(d, c_i32p))
    return r


def ntt_inverse_l4_i32(x):
    x = numpy.ascontiguousarray(x, numpy.uint64)
    r = numpy.empty(1024, numpy.uint32)
    lib().emu_ntt_inverse_l4_i32(_p(r, c_u32p), _p(x, c_u64p))
    return r


def bk_from_reference(bk):
    bk = numpy.ascontiguousarray(bk, numpy.uint64)
    out = numpy.empty_like(bk)
    lib().emu_bk_from_reference(_p(out, c_u64p), _p(bk, c_u64p), ctypes.c_long(bk.size // 1024))
    return out


def bk_to_half(bk_internal):
    bk_internal = numpy.ascontiguousarray(bk_internal, numpy.uint64)
    out = numpy.empty_like(bk_internal)
    lib().emu_bk_to_half(_p(out, c_u64p), _p(bk_internal, c_u64p), ctypes.c_long(bk_internal.size // 1024))
    return out


def bootstrap_bit(bk_internal, n, src0, p0, src1, p1, c0, mu, mask_size=1, team=False, pair=False, ring=False, team8=False):
    """src = (a [n], b scalar array [1]); returns (ext_a [1024 * mask_size], ext_b).
    team=True runs the 4-wave (k = 2: 3-wave) team variant of the body, pair=True the 2-wave variant (k = 1)."""
    a0 = numpy.ascontiguousarray(src0[0], numpy.int32); b0 = numpy.ascontiguousarray(src0[1], numpy.int32).reshape(1)
    a1 = numpy.ascontiguousarray(src1[0], numpy.int32); b1 = numpy.ascontiguousarray(src1[1], numpy.int32).reshape(1)
    out_a = numpy.empty(1024 * mask_size, numpy.int32); out_b = numpy.empty(1, numpy.int32)
    fn = lib().emu_bootstrap_bit if mask_size == 1 else lib().emu_bootstrap_bit_k2
    if team:
        fn = lib().emu_bootstrap_bit_team if mask_size == 1 else lib().emu_bootstrap_bit_team_k2
    if pair:
        fn = lib().emu_bootstrap_bit_pair
    if ring:
        fn = lib().emu_bootstrap_bit_ring_k2          # mask_size 2 only
    if team8:
        fn = lib().emu_bootstrap_bit_team8            # bk_internal must be in the half-ring layout (bk_to_half)
    fn(_p(out_a, c_i32p), _p(out_b, c_i32p), _p(bk_internal, c_u64p), ctypes.c_int(n),
                            _p(a0, c_i32p), _p(b0, c_i32p), ctypes.c_int32(p0),
                            _p(a1, c_i32p), _p(b1, c_i32p), ctypes.c_int32(p1),
                            ctypes.c_int32(c0), ctypes.c_int32(mu))
    return out_a, out_b[0]


c_f64p = ctypes.POINTER(ctypes.c_double)


def fft_forward(x):
    x = numpy.ascontiguousarray(x, numpy.int32)
    out = numpy.empty(1024, numpy.float64)
    lib().emu_fft_forward(_p(out, c_f64p), _p(x, c_i32p))
    return out.view(numpy.complex128)


def fft_inverse(x):
    x = numpy.ascontiguousarray(x, numpy.complex128).view(numpy.float64)
    out = numpy.empty(1024, numpy.int32)
    lib().emu_fft_inverse(_p(out, c_i32p), _p(x, c_f64p))
    return out


def bkf_from_reference(bk):
    bk = numpy.ascontiguousarray(bk, numpy.complex128)
    out = numpy.empty_like(bk)
    lib().emu_bkf_from_reference(_p(out.view(numpy.float64), c_f64p), _p(bk.view(numpy.float64), c_f64p),
                                 ctypes.c_long(bk.size // 512))
    return out


def bootstrap_bit_fft(bk_internal, n, src0, p0, src1, p1, c0, mu, team=False, mask_size=1, pair=False, ring=False, quad=False):
    a0 = numpy.ascontiguousarray(src0[0], numpy.int32); b0 = numpy.ascontiguousarray(src0[1], numpy.int32).reshape(1)
    a1 = numpy.ascontiguousarray(src1[0], numpy.int32); b1 = numpy.ascontiguousarray(src1[1], numpy.int32).reshape(1)
    out_a = numpy.empty(1024 * mask_size, numpy.int32); out_b = numpy.empty(1, numpy.int32)
    fn = lib().emu_bootstrap_bit_fft_team if team else lib().emu_bootstrap_bit_fft
    if pair:
        fn = lib().emu_bootstrap_bit_fft_pair
    if quad:
        fn = lib().emu_bootstrap_bit_fft_quad
    if mask_size == 2:
        fn = lib().emu_bootstrap_bit_fft_team_k2 if team else lib().emu_bootstrap_bit_fft_k2
        if ring:
            fn = lib().emu_bootstrap_bit_fft_ring_k2
        if quad:
            fn = lib().emu_bootstrap_bit_fft_hex_k2
    fn(_p(out_a, c_i32p), _p(out_b, c_i32p), _p(bk_internal.view(numpy.float64), c_f64p),
                                ctypes.c_int(n), _p(a0, c_i32p), _p(b0, c_i32p), ctypes.c_int32(p0),
                                _p(a1, c_i32p), _p(b1, c_i32p), ctypes.c_int32(p1),
                                ctypes.c_int32(c0), ctypes.c_int32(mu))
    return out_a, out_b[0]


def fft_margin():
    """(largest distance from an integer, largest magnitude) of the values rounded by the FFT inverse
    transforms since the last call (host build statistics)."""
    a = ctypes.c_double(); b = ctypes.c_double()
    lib().emu_fft_margin(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def fft_round(v):
    v = numpy.ascontiguousarray(v, numpy.float64)
    r = numpy.empty(v.shape, numpy.uint32)
    lib().emu_fft_round(_p(r, c_u32p), v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_long(v.size))
    return r


def nth_forward_small(d):
    d = numpy.ascontiguousarray(d, numpy.int32)
    r = numpy.empty(1024, numpy.uint64)
    lib().emu_nth_forward_small(_p(r, c_u64p), _p(d, c_i32p))
    return r


def nth_inverse_i32(x):
    x = numpy.ascontiguousarray(x, numpy.uint64)
    r = numpy.empty(1024, numpy.uint32)
    lib().emu_nth_inverse_i32(_p(r, c_u32p), _p(x, c_u64p))
    return r


# ---- exact-FFT engine (csrc/blind_rotate_xfft.h) ----

def bkx_from_coeffs(tgsw):
    """int32 TGSW polynomials [..., 1024] -> split key image complex128 [polys, 2, 512] in the wave layout"""
    tgsw = numpy.ascontiguousarray(tgsw, numpy.int32)
    polys = tgsw.size // 1024
    out = numpy.empty((polys, 2, 512), numpy.complex128)
    lib().emu_bkx_from_coeffs(_p(out.view(numpy.float64), c_f64p), _p(tgsw, c_i32p), ctypes.c_long(polys))
    return out


def xfft_external_product(T, row):
    """T int32 [2, 1024], row = bkx_from_coeffs of one TGSW row (8 polynomials) -> int32 [2, 1024]"""
    T = numpy.ascontiguousarray(T, numpy.int32)
    row = numpy.ascontiguousarray(row, numpy.complex128)
    assert T.shape == (2, 1024) and row.size == 8 * 2 * 512
    res = numpy.empty((2, 1024), numpy.int32)
    lib().emu_xfft_external_product(_p(res, c_i32p), _p(T, c_i32p), _p(row.view(numpy.float64), c_f64p))
    return res


def xfft_margin():
    a = ctypes.c_double(); b = ctypes.c_double()
    lib().emu_xfft_margin(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def bootstrap_bit_xfft(bkx, n, src0, p0, src1, p1, c0, mu):
    a0 = numpy.ascontiguousarray(src0[0], numpy.int32); b0 = numpy.ascontiguousarray(src0[1], numpy.int32).reshape(1)
    a1 = numpy.ascontiguousarray(src1[0], numpy.int32); b1 = numpy.ascontiguousarray(src1[1], numpy.int32).reshape(1)
    out_a = numpy.empty(1024, numpy.int32); out_b = numpy.empty(1, numpy.int32)
    bkx = numpy.ascontiguousarray(bkx, numpy.complex128)
    lib().emu_bootstrap_bit_xfft(_p(out_a, c_i32p), _p(out_b, c_i32p), _p(bkx.view(numpy.float64), c_f64p),
                                 ctypes.c_int(n), _p(a0, c_i32p), _p(b0, c_i32p), ctypes.c_int32(p0),
                                 _p(a1, c_i32p), _p(b1, c_i32p), ctypes.c_int32(p1),
                                 ctypes.c_int32(c0), ctypes.c_int32(mu))
    return out_a, out_b[0]


def xfft_external_product_k2(T, row):
    """tlwe_mask_size = 2: T int32 [3, 1024], row = bkx_from_coeffs of one TGSW row (18 polynomials) -> int32 [3, 1024]"""
    T = numpy.ascontiguousarray(T, numpy.int32)
    row = numpy.ascontiguousarray(row, numpy.complex128)
    assert T.shape == (3, 1024) and row.size == 18 * 2 * 512
    res = numpy.empty((3, 1024), numpy.int32)
    lib().emu_xfft_external_product_k2(_p(res, c_i32p), _p(T, c_i32p), _p(row.view(numpy.float64), c_f64p))
    return res


def bootstrap_bit_xfft_k2(bkx, n, src0, p0, src1, p1, c0, mu):
    a0 = numpy.ascontiguousarray(src0[0], numpy.int32); b0 = numpy.ascontiguousarray(src0[1], numpy.int32).reshape(1)
    a1 = numpy.ascontiguousarray(src1[0], numpy.int32); b1 = numpy.ascontiguousarray(src1[1], numpy.int32).reshape(1)
    out_a = numpy.empty(2048, numpy.int32); out_b = numpy.empty(1, numpy.int32)
    bkx = numpy.ascontiguousarray(bkx, numpy.complex128)
    lib().emu_bootstrap_bit_xfft_k2(_p(out_a, c_i32p), _p(out_b, c_i32p), _p(bkx.view(numpy.float64), c_f64p),
                                    ctypes.c_int(n), _p(a0, c_i32p), _p(b0, c_i32p), ctypes.c_int32(p0),
                                    _p(a1, c_i32p), _p(b1, c_i32p), ctypes.c_int32(p1),
                                    ctypes.c_int32(c0), ctypes.c_int32(mu))
    return out_a, out_b[0]


def bootstrap_bit_xfft_quad(bkx, n, src0, p0, src1, p1, c0, mu, split=False):
    """k = 1, four waves per bit (brxq_*): the same outputs as bootstrap_bit_xfft"""
    a0 = numpy.ascontiguousarray(src0[0], numpy.int32); b0 = numpy.ascontiguousarray(src0[1], numpy.int32).reshape(1)
    a1 = numpy.ascontiguousarray(src1[0], numpy.int32); b1 = numpy.ascontiguousarray(src1[1], numpy.int32).reshape(1)
    out_a = numpy.empty(1024, numpy.int32); out_b = numpy.empty(1, numpy.int32)
    bkx = numpy.ascontiguousarray(bkx, numpy.complex128)
    lib().emu_bootstrap_bit_xfft_quad(_p(out_a, c_i32p), _p(out_b, c_i32p), _p(bkx.view(numpy.float64), c_f64p),
                                      ctypes.c_int(n), _p(a0, c_i32p), _p(b0, c_i32p), ctypes.c_int32(p0),
                                      _p(a1, c_i32p), _p(b1, c_i32p), ctypes.c_int32(p1),
                                      ctypes.c_int32(c0), ctypes.c_int32(mu), ctypes.c_int(1 if split else 0))
    return out_a, out_b[0]


def bootstrap_bit_xfft_hex_k2(bkx, n, src0, p0, src1, p1, c0, mu):
    """tlwe_mask_size = 2, six waves per bit (brxq_* with K = 2): the same outputs as bootstrap_bit_xfft_k2"""
    a0 = numpy.ascontiguousarray(src0[0], numpy.int32); b0 = numpy.ascontiguousarray(src0[1], numpy.int32).reshape(1)
    a1 = numpy.ascontiguousarray(src1[0], numpy.int32); b1 = numpy.ascontiguousarray(src1[1], numpy.int32).reshape(1)
    out_a = numpy.empty(2048, numpy.int32); out_b = numpy.empty(1, numpy.int32)
    bkx = numpy.ascontiguousarray(bkx, numpy.complex128)
    lib().emu_bootstrap_bit_xfft_hex_k2(_p(out_a, c_i32p), _p(out_b, c_i32p), _p(bkx.view(numpy.float64), c_f64p),
                                        ctypes.c_int(n), _p(a0, c_i32p), _p(b0, c_i32p), ctypes.c_int32(p0),
                                        _p(a1, c_i32p), _p(b1, c_i32p), ctypes.c_int32(p1),
                                        ctypes.c_int32(c0), ctypes.c_int32(mu))
    return out_a, out_b[0]
