"""
Torus polynomial containers and the negacyclic monomial products of the step-by-step bootstrap driver
(reference: nufhe/polynomials.py:30-107; device kernel ShiftTorusPolynomial polynomials_gpu.py:31-86,
CPU reference polynomials_cpu.py:25-59).  The fused kernels do these rotations inside the blind rotation;
the functions here exist for callers of the reference's low-level API and for the step-by-step driver
(`PerformanceParameters(single_kernel_bootstrap=False)`, nufhe_amd/bootstrap.py).
"""

import pickle

import numpy

from . import _lib
from .device import ptr, int32_operand
from .numeric_functions import Torus32
from .utils import arrays_equal


def transformed_dtype(transform_type):
    """Element type of a transformed polynomial: residues mod 2^64 - 2^32 + 1 as uint64 (NTT,
    polynomial_transform_ntt.py:33-34) or complex128 (FFT, polynomial_transform_fft.py:34-35)."""
    return {'NTT': numpy.dtype('uint64'), 'FFT': numpy.dtype('complex128')}[transform_type]


def transformed_length(transform_type, polynomial_degree: int):
    """N values for the NTT, N / 2 for the folded FFT (polynomial_transform_ntt.py:45-46, _fft.py:46-47)."""
    return {'NTT': polynomial_degree, 'FFT': polynomial_degree // 2}[transform_type]


class IntPolynomialArray:
    """nufhe/polynomials.py:30-40: integer polynomials (secret-key material), ``coeffs [..., N]``."""

    def __init__(self, coeffs):
        self.coeffs = coeffs
        self.polynomial_degree = coeffs.shape[-1]
        self.shape = tuple(coeffs.shape[:-1])


class TorusPolynomialArray:
    """nufhe/polynomials.py:40-50: int32 device array ``coeffs [shape..., N]``."""

    def __init__(self, coeffs):
        import torch
        if coeffs.dtype != torch.int32:
            raise TypeError("torus polynomials are int32, got %s" % (coeffs.dtype,))
        self.coeffs = coeffs
        self.shape = tuple(coeffs.shape[:-1])
        self.polynomial_degree = coeffs.shape[-1]

    @classmethod
    def empty(cls, thr, polynomial_degree: int, shape):
        return cls(thr.array(tuple(shape) + (polynomial_degree,), Torus32))


class TransformedPolynomialArray:
    """Polynomials in the transformed domain in the REFERENCE's element order and scaling (natural frequency order; for
    the NTT the Montgomery-prepared residues `tlwe_transform_samples` produces): ``coeffs [shape..., N or N / 2]``
    (nufhe/polynomials.py:54-86).  The bootstrapping key the kernels read is a different object (library layout
    behind a key handle, `tgsw.TransformedTGswSampleArray`); this container is the exchange and storage form."""

    def __init__(self, transform_type, coeffs):
        import torch
        want = transformed_dtype(transform_type)
        have = {torch.complex128: 'complex128', torch.int64: 'uint64'}.get(coeffs.dtype, str(coeffs.dtype).split('.')[-1])
        if have != want.name:
            raise TypeError("%s polynomials are %s, got %s" % (transform_type, want.name, coeffs.dtype))
        self.transform_type = transform_type
        self.coeffs = coeffs
        self.shape = tuple(coeffs.shape[:-1])
        self.polynomial_degree = coeffs.shape[-1]

    @classmethod
    def empty(cls, thr, transform_type, polynomial_degree: int, shape):
        length = transformed_length(transform_type, polynomial_degree)
        return cls(transform_type, thr.array(tuple(shape) + (length,), transformed_dtype(transform_type)))

    def _host(self):
        host = self.coeffs.detach().cpu().numpy()
        return host.view(numpy.uint64) if host.dtype == numpy.int64 else host

    def dump(self, file_obj):
        pickle.dump(self.transform_type, file_obj)
        pickle.dump(self._host(), file_obj)

    @classmethod
    def load(cls, file_obj, thr):
        transform_type = pickle.load(file_obj)
        coeffs = pickle.load(file_obj)
        return cls(transform_type, thr.to_device(coeffs))

    def __eq__(self, other):
        return (self.__class__ == other.__class__ and self.transform_type == other.transform_type
                and arrays_equal(self._host(), other._host()))


def _shift(thr, result: TorusPolynomialArray, source: TorusPolynomialArray, powers, power_idx, minus_one, invert):
    """One launch of the monomial product: every polynomial of batch element ``i`` is multiplied by
    X^p (or X^p - 1, or X^(2N - p)), p = powers[i, power_idx]."""
    if result.polynomial_degree != 1024 or source.polynomial_degree != 1024:
        raise ValueError("the gfx950 kernels support N=1024")
    if tuple(result.coeffs.shape) != tuple(source.coeffs.shape):
        raise ValueError("result and source shapes differ: %s vs %s" % (tuple(result.coeffs.shape), tuple(source.coeffs.shape)))
    if result.coeffs.data_ptr() == source.coeffs.data_ptr():
        raise ValueError("the monomial product does not work in place")
    powers_shape = tuple(powers.shape)
    powers_view = power_idx is not None
    batch_shape = powers_shape[:-1] if powers_view else powers_shape
    stride = powers_shape[-1] if powers_view else 1
    if result.shape[:len(batch_shape)] != batch_shape:
        raise ValueError("powers of shape %s do not match polynomials of shape %s" % (powers_shape, result.shape))
    if powers_view and not (0 <= power_idx < stride):
        raise ValueError("power index %d out of range [0, %d)" % (power_idx, stride))
    batch = int(numpy.prod(batch_shape)) if batch_shape else 1
    polys = int(numpy.prod(result.shape[len(batch_shape):])) if len(result.shape) > len(batch_shape) else 1
    thr.check_stream()
    powers = int32_operand("powers", powers, thr.device)
    src = source.coeffs.contiguous()
    out = result.coeffs if result.coeffs.is_contiguous() else result.coeffs.new_empty(result.coeffs.shape)
    _lib.call("nufhe_shift_torus_polynomial", thr.handle, ptr(out), ptr(src), ptr(powers), stride,
              power_idx if powers_view else 0, batch, polys, int(minus_one), int(invert))
    if out is not result.coeffs:
        result.coeffs.copy_(out)


def shift_tp_inverted_power(thr, result: TorusPolynomialArray, powers, source: TorusPolynomialArray):
    """result = X^(2N - powers) * source, one power per polynomial (nufhe/polynomials.py:90-95)."""
    _shift(thr, result, source, powers, None, minus_one=False, invert=True)


def shift_tp_minus_one_power_from_array(thr, result: TorusPolynomialArray, powers, power_idx: int,
                                        source: TorusPolynomialArray):
    """result = (X^powers[..., power_idx] - 1) * source; the trailing axes of ``result`` beyond the shape of
    ``powers[..., 0]`` share a power (nufhe/polynomials.py:99-104)."""
    _shift(thr, result, source, powers, int(power_idx), minus_one=True, invert=False)
