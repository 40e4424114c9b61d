"""
TLWE parameter / key records and the device side of TLWE key material
(reference: nufhe/tlwe.py:48-207, nufhe/polynomials.py:30-86).
"""

import numpy

from . import _lib
from .device import ptr
from .lwe import LweParams
from .numeric_functions import Torus32
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


class IntPolynomialArray:
    """nufhe/polynomials.py:30-40"""

    def __init__(self, coeffs):
        self.coeffs = coeffs
        self.polynomial_degree = coeffs.shape[-1]
        self.shape = tuple(coeffs.shape[:-1])


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


def tlwe_encrypt_zero(thr, rng, shape, noise: float, key: TLweKey):
    """
    Homogeneous TLWE samples of zero (nufhe/tlwe.py:185-196, TLweEncryptZero tlwe_gpu.py:111-196):
    returns an int32 device array ``shape + (k + 1, N)``.  Randomness is drawn on the host in the
    reference's order (uniform mask first, then the Gaussian body noise); the polynomial products
    run on the GPU (forward NTT x forward NTT -> pointwise product -> inverse NTT).
    """
    params = key.params
    if params.polynomial_degree != 1024:
        raise ValueError("the gfx950 kernels support N=1024")
    k = params.mask_size
    shape = tuple(shape)
    noises1 = rand_uniform_torus32(thr, rng, shape + (k, 1024))
    noises2 = rand_gaussian_torus32(thr, rng, 0, noise, shape + (1024,))
    batch = int(numpy.prod(shape))
    result = thr.array(shape + (k + 1, 1024), Torus32)
    _lib.call("nufhe_tlwe_encrypt_zero", thr.handle, ptr(result), ptr(key.key.coeffs.contiguous()),
              ptr(noises1), ptr(noises2), batch, k)
    return result
