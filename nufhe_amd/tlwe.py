"""
TLWE parameter / key records, the accumulator container with the per-step operations of the step-by-step
bootstrap driver, and the device side of TLWE key material (reference: nufhe/tlwe.py:48-207).
"""

import pickle

import numpy

from . import _lib
from .device import ptr
from .lwe import LweParams, LweSampleArray
from .numeric_functions import Torus32, ErrorFloat
from .polynomials import (IntPolynomialArray, TorusPolynomialArray, TransformedPolynomialArray,
                          shift_tp_minus_one_power_from_array)
from .utils import arrays_equal
from .random_numbers import rand_uniform_bool, rand_uniform_torus32, rand_gaussian_torus32


class TLweParams:
    """nufhe/tlwe.py:48-74"""

    def __init__(self, polynomial_degree: int, mask_size: int, min_noise: float, max_noise: float,
                 transform_type):
        self.polynomial_degree = polynomial_degree
        self.mask_size = mask_size
        self.min_noise = min_noise
        self.max_noise = max_noise
        self.extracted_lweparams = LweParams(polynomial_degree * mask_size, min_noise, max_noise)
        self.transform_type = transform_type

    def __eq__(self, other):
        return (
            self.__class__ == other.__class__
            and self.polynomial_degree == other.polynomial_degree
            and self.mask_size == other.mask_size
            and self.min_noise == other.min_noise
            and self.max_noise == other.max_noise
            and self.transform_type == other.transform_type)

    def __hash__(self):
        return hash((
            self.__class__, self.polynomial_degree, self.mask_size,
            self.min_noise, self.max_noise, self.transform_type))


# pickled under the reference's module path (nufhe_amd/serialization.py)
TLweParams.__module__ = 'nufhe.tlwe'


class TLweKey:
    """nufhe/tlwe.py:77-91: ``mask_size`` binary polynomials."""

    def __init__(self, params: TLweParams, key):
        self.params = params
        self.key = key

    @classmethod
    def from_rng(cls, thr, params: TLweParams, rng):
        key = IntPolynomialArray(
            rand_uniform_bool(thr, rng, (params.mask_size, params.polynomial_degree)))
        return cls(params, key)


class TLweSampleArray:
    """
    The accumulator of the blind rotation (nufhe/tlwe.py:94-112): ``a`` = TorusPolynomialArray
    ``[shape..., k + 1, N]`` (mask polynomials, then the body), ``current_variances [shape...]``.
    """

    def __init__(self, params: TLweParams, a: TorusPolynomialArray, current_variances):
        self.a = a
        self.current_variances = current_variances
        self.shape = tuple(current_variances.shape)
        self.params = params

    @classmethod
    def empty(cls, thr, params: TLweParams, shape):
        shape = tuple(shape)
        a = TorusPolynomialArray.empty(thr, params.polynomial_degree, shape + (params.mask_size + 1,))
        return cls(params, a, thr.zeros(shape, ErrorFloat))


class TransformedTLweSampleArray:
    """TLWE samples in the transformed domain, reference element order (nufhe/tlwe.py:115-153): ``a`` =
    TransformedPolynomialArray ``[shape..., k + 1, N or N / 2]``."""

    def __init__(self, params: TLweParams, a: TransformedPolynomialArray, current_variances):
        self.a = a
        self.current_variances = current_variances
        self.shape = tuple(current_variances.shape)
        self.params = params

    @classmethod
    def empty(cls, thr, params: TLweParams, shape):
        shape = tuple(shape)
        a = TransformedPolynomialArray.empty(thr, params.transform_type, params.polynomial_degree,
                                             shape + (params.mask_size + 1,))
        return cls(params, a, thr.zeros(shape, ErrorFloat))

    def dump(self, file_obj):
        pickle.dump(self.params, file_obj)
        self.a.dump(file_obj)
        pickle.dump(self.current_variances.detach().cpu().numpy(), file_obj)

    @classmethod
    def load(cls, file_obj, thr):
        params = pickle.load(file_obj)
        a = TransformedPolynomialArray.load(file_obj, thr)
        current_variances = pickle.load(file_obj)
        return cls(params, a, thr.to_device(current_variances))

    def __eq__(self, other):
        return (self.__class__ == other.__class__ and self.params == other.params and self.a == other.a
                and arrays_equal(self.current_variances, other.current_variances))


def tlwe_transform_samples(thr, result: TransformedTLweSampleArray, source: TLweSampleArray, perf_params=None):
    """result = forward transform of every polynomial of ``source`` in the reference's element order: natural-order
    negacyclic NTT followed by the Montgomery preparation x -> x 2^64 mod P, or the folded FFT-512; the variances are
    copied (nufhe/tlwe.py:199-207, TLweTransformSamples tlwe_gpu.py:199-236, transform/arithmetic.py:172-195)."""
    params = source.params
    if params.polynomial_degree != 1024:
        raise ValueError("the gfx950 kernels support N=1024")
    if result.params != params or result.shape != source.shape:
        raise ValueError("result does not match the source samples")
    thr.check_stream()
    src = source.a.coeffs.contiguous()
    out = result.a.coeffs if result.a.coeffs.is_contiguous() else result.a.coeffs.new_empty(result.a.coeffs.shape)
    batch = src.numel() // 1024
    if params.transform_type == 'NTT':
        _lib.call("nufhe_ntt_forward_i32", thr.handle, ptr(out), ptr(src), batch)
        _lib.call("nufhe_ff_op", thr.handle, ptr(out), ptr(out), None, None, None, None, 6, 64, out.numel())
    else:
        _lib.call("nufhe_fft_forward_i32", thr.handle, ptr(out), ptr(src), batch)
    if out is not result.a.coeffs:
        result.a.coeffs.copy_(out)
    result.current_variances.copy_(source.current_variances)


def tlwe_noiseless_trivial(thr, result: TLweSampleArray, mu: TorusPolynomialArray):
    """result = (0, ..., 0, mu) with variance 0 (nufhe/tlwe.py:156-158, TLweNoiselessTrivial tlwe_gpu.py:32-74,
    tlwe_cpu.py:26-38)."""
    if tuple(mu.coeffs.shape) != result.shape + (result.params.polynomial_degree,):
        raise ValueError("mu has shape %s, the samples %s" % (tuple(mu.coeffs.shape), result.shape))
    thr.check_stream()
    k = result.params.mask_size
    result.a.coeffs[..., :k, :] = 0
    result.a.coeffs[..., k, :] = mu.coeffs
    result.current_variances.zero_()


def tlwe_extract_lwe_samples(thr, result: LweSampleArray, x: TLweSampleArray):
    """LWE sample of the constant coefficient under the extracted key: a[m N + j] = A_m[0] for j = 0 and
    -A_m[N - j] otherwise, b = B[0]; the variances are left alone, as in the reference
    (nufhe/tlwe.py:161-164, tlwe_gpu.mako:54-84, tlwe_cpu.py:41-60)."""
    k = x.params.mask_size
    N = x.params.polynomial_degree
    if N != 1024:
        raise ValueError("the gfx950 kernels support N=1024")
    if tuple(result.shape) != x.shape or result.a.shape[-1] != k * N:
        raise ValueError("result of shape %s x %d does not take the extraction of %s TLWE samples (k = %d)" % (
            tuple(result.shape), result.a.shape[-1], x.shape, k))
    thr.check_stream()
    batch = int(numpy.prod(x.shape)) if x.shape else 1
    src = x.a.coeffs.contiguous()
    a = result.a if result.a.is_contiguous() else result.a.new_empty(result.a.shape)
    b = result.b if result.b.is_contiguous() else result.b.new_empty(result.b.shape)
    _lib.call("nufhe_tlwe_extract", thr.handle, ptr(a), ptr(b), ptr(src), batch, k)
    if a is not result.a:
        result.a.copy_(a)
    if b is not result.b:
        result.b.copy_(b)


def tlwe_shift_polynomials(thr, result: TLweSampleArray, bk: TLweSampleArray, powers, powers_idx: int):
    """result.a = (X^powers[..., powers_idx] - 1) * bk.a, all k + 1 polynomials of a sample by the same power
    (nufhe/tlwe.py:168-169; ``bk`` is the reference's name for the SOURCE sample, kept for keyword callers)."""
    shift_tp_minus_one_power_from_array(thr, result.a, powers, powers_idx, bk.a)


def tlwe_add_to(thr, result: TLweSampleArray, source: TLweSampleArray):
    """result += source, int32 wraparound on the coefficients (nufhe/tlwe.py:173-175)."""
    thr.check_stream()
    result.a.coeffs += source.a.coeffs
    result.current_variances += source.current_variances


def tlwe_copy(thr, result: TLweSampleArray, source: TLweSampleArray):
    """result = source (nufhe/tlwe.py:179-181)."""
    thr.check_stream()
    result.a.coeffs.copy_(source.a.coeffs)
    result.current_variances.copy_(source.current_variances)


def tlwe_encrypt_zero(thr, rng, result: TLweSampleArray, noise: float, key: TLweKey, perf_params=None):
    """
    Fills ``result`` with homogeneous TLWE samples of zero, variance noise^2 (nufhe/tlwe.py:185-196, TLweEncryptZero
    tlwe_gpu.py:111-196, reference tlwe_cpu.py:64-89): a = (uniform mask, Gaussian noise + sum_i mask_i * key_i).
    Randomness is drawn on the host in the reference's order (uniform mask first, then the Gaussian body noise); the
    polynomial products run on the GPU (forward NTT x forward NTT -> pointwise product -> inverse NTT).
    """
    params = key.params
    if params.polynomial_degree != 1024:
        raise ValueError("the gfx950 kernels support N=1024")
    if result.params != params:
        raise ValueError("the samples and the key have different TLWE parameters")
    k = params.mask_size
    shape = result.shape
    noises1 = rand_uniform_torus32(thr, rng, shape + (k, 1024))
    noises2 = rand_gaussian_torus32(thr, rng, 0, noise, shape + (1024,))
    batch = int(numpy.prod(shape)) if shape else 1
    thr.check_stream()
    out = result.a.coeffs if result.a.coeffs.is_contiguous() else result.a.coeffs.new_empty(result.a.coeffs.shape)
    _lib.call("nufhe_tlwe_encrypt_zero", thr.handle, ptr(out), ptr(key.key.coeffs.contiguous()),
              ptr(noises1), ptr(noises2), batch, k)
    if out is not result.a.coeffs:
        result.a.coeffs.copy_(out)
    result.current_variances.fill_(float(noise) ** 2)
