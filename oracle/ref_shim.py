"""
ref_shim -- load the reference's own CPU reference functions (nucypher/nufhe ``*_cpu.py``)
in a container that has neither ``reikna`` nor a GPU.  TEST INFRASTRUCTURE ONLY.

The reference package cannot be imported as ``nufhe`` (almost every module imports
``reikna`` at the top).  The per-kernel NumPy reference functions, however, only need

* ``reikna.helpers.product`` / ``min_blocks``      (tgsw_cpu.py:20, tlwe_cpu.py:20, polynomials_cpu.py:20)
* ``nufhe.numeric_functions.{Torus32, Int32, ErrorFloat, double_to_t32}``
* ``nufhe.polynomial_transform.get_transform``     (tgsw_cpu.py:23, tlwe_cpu.py:23)

so this module injects a stub ``reikna.helpers`` and a synthetic package ``_nufhe_ref`` whose
sub-modules are the reference's *unmodified source files* executed from where they lie under
``/root/reference`` (nothing is copied into this repository).

``nufhe/transform/ntt.py`` and ``nufhe/polynomial_transform_ntt.py`` import reikna/mako at module
top, so their few lines of pure-NumPy logic are restated below (``NttTransformRef``), on top of
the reference's own ``transform/ntt_cpu.py`` (which loads standalone):

* ``ntt_transform_ref``                      nufhe/transform/ntt.py:30-44
* ``forward/inverse_transform_ref``          nufhe/polynomial_transform_ntt.py:45-50
* ``transformed_space_*_ref``                nufhe/polynomial_transform_ntt.py:53-69

One NumPy-2 hazard is patched: ``ntt_cpu._gnum_to_i32`` uses
``numpy.int32(val & 0xffffffff)`` which raises OverflowError on NumPy >= 2 for values >= 2^31;
the intended behaviour (NumPy 1.x silent wrap) is restored explicitly.
"""

import importlib.util
import os
import sys
import types

import numpy

REFERENCE_ROOT = os.environ.get("NUFHE_REFERENCE_ROOT", "/root/reference")
PKG = "_nufhe_ref"


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "nufhe", "tgsw_cpu.py"))


def _load(name, relpath):
    fullname = PKG + "." + name
    if fullname in sys.modules:
        return sys.modules[fullname]
    spec = importlib.util.spec_from_file_location(
        fullname, os.path.join(REFERENCE_ROOT, "nufhe", relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[fullname] = mod
    spec.loader.exec_module(mod)
    return mod


class NttTransformRef:
    """Restatement of the pure-NumPy part of polynomial_transform_ntt.py / transform/ntt.py."""

    def __init__(self, ntt_cpu):
        self.ntt_cpu = ntt_cpu

    @staticmethod
    def transformed_dtype():
        return numpy.dtype('uint64')                      # polynomial_transform_ntt.py:29-30

    @staticmethod
    def transformed_length(N):
        return N                                          # polynomial_transform_ntt.py:41-42

    def ntt_transform_ref(self, data, inverse=False, i32_conversion=False):
        ntt_cpu = self.ntt_cpu                            # transform/ntt.py:30-44
        N = data.shape[-1]
        data = ntt_cpu.gnum(data)
        w = ntt_cpu.root_of_unity(2 * N)
        forward_coeffs = numpy.array([w**int(i) for i in numpy.arange(N)])
        if inverse:
            inverse_coeffs = ntt_cpu.gnum(1) / forward_coeffs
            res = ntt_cpu.ntt(data, True) * inverse_coeffs
            if i32_conversion:
                return ntt_cpu.gnum_to_i32(res)
            return ntt_cpu.gnum_to_u64(res)
        return ntt_cpu.gnum_to_u64(ntt_cpu.ntt(data * forward_coeffs, False))

    def forward_transform_ref(self, data):
        return self.ntt_transform_ref(data, i32_conversion=True)

    def inverse_transform_ref(self, data):
        return self.ntt_transform_ref(data, i32_conversion=True, inverse=True)

    def transformed_space_add_ref(self, d1, d2):
        g = self.ntt_cpu
        return g.gnum_to_u64(g.gnum(d1) + g.gnum(d2))

    def transformed_space_mul_ref(self, d1, d2):
        g = self.ntt_cpu
        return g.gnum_to_u64(g.gnum(d1) * g.gnum(d2))

    def transformed_space_mul_prepared_ref(self, d1, d2):
        g = self.ntt_cpu
        coeff = g.gnum(0xfffffffe00000001)                # polynomial_transform_ntt.py:66
        return g.gnum_to_u64(g.gnum(d1) * g.gnum(d2) * coeff)


class FftTransformRef:
    """Restatement of the pure-NumPy part of polynomial_transform_fft.py / transform/fft.py
    (both import reikna at module top).  numpy.fft is the reference's own engine here."""

    @staticmethod
    def transformed_dtype():
        return numpy.dtype('complex128')                  # polynomial_transform_fft.py:28-29

    @staticmethod
    def transformed_length(N):
        return N // 2                                     # polynomial_transform_fft.py:51-52

    @staticmethod
    def fft_transform_ref(data, inverse=False, i32_conversion=False):
        if i32_conversion and not inverse:                # transform/fft.py:27-51
            N = data.shape[-1]
        else:
            N = data.shape[-1] * 2
        batch_shape = data.shape[:-1]
        data = data.reshape(int(numpy.prod(batch_shape, dtype=numpy.int64)), data.shape[-1])
        coeffs = numpy.exp(-2j * numpy.pi * numpy.arange(N // 2) / N / 2)
        f64_to_i32 = lambda x: numpy.round(x).astype(numpy.int64).astype(numpy.int32)
        if inverse:
            final_shape = batch_shape + ((N,) if i32_conversion else (N // 2,))
            res = numpy.fft.ifft(data).conj() * coeffs
            if i32_conversion:
                res = numpy.concatenate([f64_to_i32(res.real), f64_to_i32(res.imag)], axis=1)
            return res.reshape(final_shape)
        else:
            if i32_conversion:
                data = (data[:, :N // 2] - 1j * data[:, N // 2:])
            return numpy.fft.fft(data * coeffs).reshape(batch_shape + (N // 2,))

    def forward_transform_ref(self, data):
        return self.fft_transform_ref(data, i32_conversion=True)

    def inverse_transform_ref(self, data):
        return self.fft_transform_ref(data, i32_conversion=True, inverse=True)

    @staticmethod
    def transformed_space_add_ref(d1, d2):
        return d1 + d2

    @staticmethod
    def transformed_space_mul_ref(d1, d2):
        return d1 * d2

    @staticmethod
    def transformed_space_mul_prepared_ref(d1, d2):
        return d1 * d2                                    # polynomial_transform_fft.py:71-72


_loaded = None


def load():
    """Returns a namespace with the reference CPU modules:
    ntt_cpu, lwe_cpu, tlwe_cpu, tgsw_cpu, polynomials_cpu, numeric_functions_cpu, transform."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference sources not found under " + REFERENCE_ROOT)

    # stub reikna.helpers (product, min_blocks) -- the only reikna use of the *_cpu.py files
    if 'reikna' not in sys.modules:
        reikna = types.ModuleType('reikna')
        helpers = types.ModuleType('reikna.helpers')
        helpers.product = lambda seq: int(numpy.prod(list(seq), dtype=numpy.int64)) if len(seq) else 1
        helpers.min_blocks = lambda length, block: (length - 1) // block + 1
        reikna.helpers = helpers
        sys.modules['reikna'] = reikna
        sys.modules['reikna.helpers'] = helpers

    pkg = types.ModuleType(PKG)
    pkg.__path__ = []
    sys.modules[PKG] = pkg

    ntt_cpu = _load('ntt_cpu', os.path.join('transform', 'ntt_cpu.py'))

    # NumPy >= 2 hazard: restore the two's-complement wrap of ntt_cpu.py:74-80
    def _gnum_to_i32(x):
        med = x.modulus // 2
        val = x.val
        return numpy.uint32(((val & 0xffffffff) - (1 if val > med else 0)) & 0xffffffff).astype(numpy.int32)
    ntt_cpu._gnum_to_i32 = _gnum_to_i32
    ntt_cpu.gnum_to_i32 = numpy.vectorize(_gnum_to_i32, otypes=[numpy.int32])
    ntt_cpu.gnum_to_u64 = numpy.vectorize(lambda x: numpy.uint64(x.val), otypes=[numpy.uint64])

    transform = NttTransformRef(ntt_cpu)
    fft_transform = FftTransformRef()

    # synthetic nufhe.numeric_functions / numeric_functions_gpu (dtypes + double_to_t32)
    nf = types.ModuleType(PKG + '.numeric_functions')
    nf.Torus32 = numpy.int32                              # numeric_functions_gpu.py:30-36
    nf.Int32 = numpy.int32
    nf.ErrorFloat = numpy.float32
    nf.double_to_t32 = lambda d: ((d - numpy.trunc(d)) * 2**32).astype(numpy.int32)  # numeric_functions.py:39-40
    sys.modules[PKG + '.numeric_functions'] = nf
    sys.modules[PKG + '.numeric_functions_gpu'] = nf
    pkg.numeric_functions = nf

    pt = types.ModuleType(PKG + '.polynomial_transform')
    def get_transform(transform_type):
        assert transform_type in ('NTT', 'FFT')
        return transform if transform_type == 'NTT' else fft_transform
    pt.get_transform = get_transform
    sys.modules[PKG + '.polynomial_transform'] = pt

    ns = types.SimpleNamespace(
        ntt_cpu=ntt_cpu,
        transform=transform,
        fft_transform=fft_transform,
        numeric_functions_cpu=_load('numeric_functions_cpu', 'numeric_functions_cpu.py'),
        polynomials_cpu=_load('polynomials_cpu', 'polynomials_cpu.py'),
        lwe_cpu=_load('lwe_cpu', 'lwe_cpu.py'),
        tlwe_cpu=_load('tlwe_cpu', 'tlwe_cpu.py'),
        tgsw_cpu=_load('tgsw_cpu', 'tgsw_cpu.py'),
        )
    _loaded = ns
    return ns


class RefTLweParams:
    """Minimal stand-ins for the parameter records the reference functions read
    (nufhe/tlwe.py:48-62, nufhe/tgsw.py:43-57)."""

    def __init__(self, polynomial_degree=1024, mask_size=1, transform_type='NTT'):
        self.polynomial_degree = polynomial_degree
        self.mask_size = mask_size
        self.transform_type = transform_type


class RefTGswParams:

    def __init__(self, tlwe_params, decomp_length=2, bs_log2_base=10):
        decomp_range = numpy.arange(1, decomp_length + 1)
        self.base_powers = (2**(32 - decomp_range * bs_log2_base)).astype(numpy.int32)      # tgsw.py:47
        self.offset = (
            self.base_powers.astype(numpy.int64).sum() * (2**bs_log2_base // 2)).astype(numpy.int32)  # :50-52
        self.decomp_length = decomp_length
        self.bs_log2_base = bs_log2_base
        self.tlwe_params = tlwe_params
