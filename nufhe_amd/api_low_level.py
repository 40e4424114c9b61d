"""
Low-level API: scheme parameters, keys, encrypt/decrypt (reference: nufhe/api_low_level.py:31-302).
"""

import io
import pickle

import numpy

from .bootstrap import BootstrapKey, NativeCloudKey
from .lwe import LweParams, LweKey, LweSampleArray, lwe_encrypt, lwe_decrypt, LweKeyswitchKey
from .numeric_functions import phase_to_t32
from .performance import PerformanceParameters
from .tgsw import TGswParams, TGswKey
from .tlwe import TLweParams


class NuFHEParameters:
    """
    Parameters of the FHE scheme (nufhe/api_low_level.py:31-87); the defaults correspond to about
    128 bits of security.

    :param transform_type: ``'NTT'`` (64-bit-prime number-theoretic transform, bit-exact) or
        ``'FFT'`` (fp64 folded FFT, faster; results equal the exact ones unless an fp64 rounding
        error reaches half a torus LSB -- see DESIGN.md for the stated tolerance).
    """

    #: the one parameter set of the scheme (the values of nufhe/api_low_level.py:49-61); standard
    #: deviations are given as multiples of sqrt(2 / pi)
    SCHEME = {
        "lwe_size": 500,                      # n: size of the in/out LWE samples
        "tlwe_polynomial_degree": 1024,       # N
        "bootstrap_decomposition": (2, 10),   # (l, log2 Bg) of the gadget
        "keyswitch_decomposition": (8, 2),    # (t, log2 base)
        "stdev_keyswitch": 2.0**-15,          # noise of fresh / keyswitched samples
        "stdev_bootstrap": 9e-9,              # noise of the bootstrapping-key samples
        "stdev_max": 2.0**-6,                 # largest tolerated noise: a quarter of the 1/16 margin
    }
    #: (transform, mask size) pairs with a gfx950 bootstrap kernel
    SUPPORTED = {('NTT', 1), ('NTT', 2), ('FFT', 1), ('FFT', 2)}

    def __init__(self, transform_type='NTT', tlwe_mask_size=1):
        assert transform_type in ('FFT', 'NTT')
        assert tlwe_mask_size >= 1
        if (transform_type, tlwe_mask_size) not in self.SUPPORTED:
            raise NotImplementedError("no gfx950 kernel for transform %s with tlwe_mask_size=%d (have: %s)" % (
                transform_type, tlwe_mask_size, sorted(self.SUPPORTED)))
        cfg = self.SCHEME
        unit = float(numpy.sqrt(2 / numpy.pi))
        accumulator = TLweParams(cfg["tlwe_polynomial_degree"], tlwe_mask_size, cfg["stdev_bootstrap"] * unit,
                                 cfg["stdev_max"] * unit, transform_type)
        self.ks_decomp_length, self.ks_log2_base = cfg["keyswitch_decomposition"]
        self.in_out_params = LweParams(cfg["lwe_size"], cfg["stdev_keyswitch"] * unit, cfg["stdev_max"] * unit)
        self.tgsw_params = TGswParams(accumulator, *cfg["bootstrap_decomposition"])
        self._transform_type = transform_type
        self._tlwe_mask_size = tlwe_mask_size

    def __hash__(self):
        return hash((self.__class__, self._transform_type, self._tlwe_mask_size))

    def __eq__(self, other):
        return (
            self.__class__ == other.__class__
            and self._transform_type == other._transform_type
            and self._tlwe_mask_size == other._tlwe_mask_size)


# pickled under the reference's module path (nufhe_amd/serialization.py)
NuFHEParameters.__module__ = 'nufhe.api_low_level'


class NuFHESecretKey:
    """A secret key (nufhe/api_low_level.py:90-148)."""

    def __init__(self, params: NuFHEParameters, lwe_key: LweKey):
        self.params = params
        self.lwe_key = lwe_key

    @classmethod
    def from_rng(cls, thr, params: NuFHEParameters, rng):
        return cls(params, LweKey.from_rng(thr, params.in_out_params, rng))

    def dump(self, file_obj):
        pickle.dump(self.params, file_obj)
        self.lwe_key.dump(file_obj)

    def dumps(self):
        file_obj = io.BytesIO()
        self.dump(file_obj)
        return file_obj.getvalue()

    @classmethod
    def load(cls, file_obj, thr):
        params = pickle.load(file_obj)
        lwe_key = LweKey.load(file_obj, thr)
        return cls(params, lwe_key)

    @classmethod
    def loads(cls, s: bytes, thr):
        return cls.load(io.BytesIO(s), thr)

    def __eq__(self, other):
        return (
            self.__class__ == other.__class__
            and self.params == other.params
            and self.lwe_key == other.lwe_key)


class NuFHECloudKey:
    """A cloud key: bootstrapping key + keyswitch key (nufhe/api_low_level.py:151-239)."""

    def __init__(self, params: NuFHEParameters, bootstrap_key: BootstrapKey,
                 keyswitch_key: LweKeyswitchKey, native: NativeCloudKey):
        self.params = params
        self.bootstrap_key = bootstrap_key
        self.keyswitch_key = keyswitch_key
        self._native = native

    @staticmethod
    def _attach_keyswitch(native, ks: LweKeyswitchKey):
        import ctypes
        from . import _lib
        lwe = ks.lwe
        # the library stores decomposition length 8 / base 4 keys of ext_size rows (nufhe_ks_upload
        # reads exactly that many words): reject anything else before handing over host pointers
        ext_size = native.mask_size * 1024
        expect = (ext_size, 8, 4)
        if (tuple(lwe.shape) != expect or lwe.a.shape[-1] != native.lwe_size
                or lwe.a.dtype != numpy.int32 or lwe.b.dtype != numpy.int32
                or lwe.current_variances.dtype != numpy.float32):
            raise ValueError("keyswitch key of shape %s x %d (%s) does not match the cloud key's %s x %d (int32)" % (
                tuple(lwe.shape), lwe.a.shape[-1], lwe.a.dtype, expect, native.lwe_size))
        _lib.call("nufhe_ks_upload", native.handle,
                  lwe.a.ctypes.data_as(ctypes.c_void_p), lwe.b.ctypes.data_as(ctypes.c_void_p),
                  lwe.current_variances.ctypes.data_as(ctypes.c_void_p))
        ks._native = native

    @classmethod
    def from_rng(cls, thr, params: NuFHEParameters, rng, secret_key: NuFHESecretKey, perf_params=None):
        """Generates a cloud key in the reference's RNG order (api_low_level.py:174-196):
        TGSW key, bootstrapping key, keyswitch key."""
        native = NativeCloudKey(thr, params.in_out_params.size, params._transform_type, params._tlwe_mask_size)
        tgsw_key = TGswKey.from_rng(thr, params.tgsw_params, rng)
        bk = BootstrapKey.from_rng(thr, rng, secret_key.lwe_key, tgsw_key, native, perf_params)
        ks = LweKeyswitchKey.from_tgsw_key(
            thr, rng, params.ks_decomp_length, params.ks_log2_base, secret_key.lwe_key, tgsw_key, native=native)
        return cls(params, bk, ks, native)

    def set_engine(self, engine):
        """'native' | 'exact-fft': which exact arithmetic computes this NTT key's gates (NativeCloudKey.set_engine);
        results are bit-identical, the exact-FFT engine is ~2 x faster on large batches (tlwe_mask_size 1 and 2)."""
        self._native.set_engine(engine)
        return self

    @property
    def engine(self):
        return self._native.get_engine()

    def device_image(self):
        """(params, uint8 device tensor): the key as the kernels hold it, for device-to-device replication
        (multi_gpu.broadcast_cloud_key); the serialized form for everything else is dump / dumps."""
        return self.params, self._native.export_image()

    @classmethod
    def from_device_image(cls, thr, params: NuFHEParameters, image):
        """Rebuilds a cloud key on ``thr`` from a device image made by ``device_image`` (same library build)."""
        native = NativeCloudKey(thr, params.in_out_params.size, params._transform_type, params._tlwe_mask_size)
        native.import_image(image)
        bk = BootstrapKey(params.in_out_params, params.tgsw_params, native)
        ks = LweKeyswitchKey(native=native, params=params.in_out_params,
                             shape=(params._tlwe_mask_size * params.tgsw_params.tlwe_params.polynomial_degree,
                                    params.ks_decomp_length, 2**params.ks_log2_base))
        return cls(params, bk, ks, native)

    def dump(self, file_obj):
        pickle.dump(self.params, file_obj)
        self.bootstrap_key.dump(file_obj)
        self.keyswitch_key.dump(file_obj)

    def dumps(self):
        file_obj = io.BytesIO()
        self.dump(file_obj)
        return file_obj.getvalue()

    @classmethod
    def load(cls, file_obj, thr):
        params = pickle.load(file_obj)
        native = NativeCloudKey(thr, params.in_out_params.size, params._transform_type, params._tlwe_mask_size)
        bootstrap_key = BootstrapKey.load(file_obj, thr, native)
        keyswitch_key = LweKeyswitchKey.load(file_obj, thr)
        cls._attach_keyswitch(native, keyswitch_key)
        return cls(params, bootstrap_key, keyswitch_key, native)

    @classmethod
    def loads(cls, s: bytes, thr):
        return cls.load(io.BytesIO(s), thr)

    def __eq__(self, other):
        return (
            self.__class__ == other.__class__
            and self.params == other.params
            and self.bootstrap_key == other.bootstrap_key
            and self.keyswitch_key == other.keyswitch_key)


def make_key_pair(thr, rng, **params):
    """nufhe/api_low_level.py:242-250"""
    nufhe_params = NuFHEParameters(**params)
    secret_key = NuFHESecretKey.from_rng(thr, nufhe_params, rng)
    cloud_key = NuFHECloudKey.from_rng(thr, nufhe_params, rng, secret_key)
    return secret_key, cloud_key


_1s8 = phase_to_t32(1, 8)


def bool_to_t32(bits):
    bits = numpy.asarray(bits).astype(bool)
    return numpy.where(bits, _1s8, -_1s8).astype(numpy.int32)


def t32_to_bool(mus):
    return numpy.asarray(mus) > 0


def encrypt(thr, rng, key: NuFHESecretKey, message):
    """Encrypts an array of bits (nufhe/api_low_level.py:266-281)."""
    message = numpy.asarray(message)
    result = empty_ciphertext(thr, key.params, message.shape)
    mus = bool_to_t32(message)
    noise = key.params.in_out_params.min_noise
    lwe_encrypt(thr, rng, result, mus, noise, key.lwe_key)
    return result


def decrypt(thr, key: NuFHESecretKey, ciphertext: LweSampleArray):
    """Decrypts to a numpy bool array of the ciphertext's shape (nufhe/api_low_level.py:284-295)."""
    mus = lwe_decrypt(thr, ciphertext, key.lwe_key)
    return t32_to_bool(mus)


def empty_ciphertext(thr, params: NuFHEParameters, shape):
    """nufhe/api_low_level.py:298-302"""
    return LweSampleArray.empty(thr, params.in_out_params, tuple(shape))
