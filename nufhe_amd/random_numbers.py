"""
Host random number generators (reference: nufhe/random_numbers.py:46-151).  As in the reference,
randomness is drawn on the host and uploaded; the draw ORDER of key generation and encryption
 is preserved so that one seed gives the same keys and ciphertexts.
"""

import os

import numpy

from .numeric_functions import Torus32, Int32, double_to_t32


class DeterministicRNG:
    """A fast, seedable, not cryptographically secure RNG (random_numbers.py:46-62)."""

    def __init__(self, seed=None):
        self.rng = numpy.random.RandomState(seed)

    def uniform_bool(self, shape):
        return self.rng.randint(0, 2, size=shape, dtype=Int32)

    def uniform_torus32(self, shape):
        return self.rng.randint(-2**31, 2**31, size=shape, dtype=Torus32)

    def gauss(self, shape, std_dev):
        return self.rng.normal(size=shape, scale=std_dev)


class SecureRNG:
    """
    Cryptographically secure randomness: every value is derived from ``os.urandom`` bytes (same three
    methods as the reference's SecureRNG, random_numbers.py:65-124, so the two are interchangeable in
    ``Context(rng=...)``; the construction below is this package's own).

    * bits: the low bit of one random byte each;
    * torus elements: 4 random bytes each, read as a little-endian int32;
    * Gaussians: Marsaglia's polar method on pairs of uniforms in (-1, 1) built from 53 random bits;
      rejected pairs (about 21 %) are redrawn until the request is filled.
    """

    def _bytes(self, count, dtype):
        dtype = numpy.dtype(dtype)
        return numpy.frombuffer(os.urandom(int(count) * dtype.itemsize), dtype=dtype)

    def uniform_bool(self, shape):
        n = int(numpy.prod(shape, dtype=numpy.int64))
        return (self._bytes(n, numpy.uint8) & 1).astype(Int32).reshape(shape)

    def uniform_torus32(self, shape):
        n = int(numpy.prod(shape, dtype=numpy.int64))
        return self._bytes(n, '<i4').astype(Torus32).reshape(shape)

    def _symmetric_uniform(self, n):
        # 53 random bits -> k / 2^53 in [0, 1) -> 2 x - 1 in [-1, 1)
        k = self._bytes(n, '<u8') >> numpy.uint64(11)
        return k.astype(numpy.float64) * (2.0 / 9007199254740992.0) - 1.0

    def gauss(self, shape, std_dev):
        n = int(numpy.prod(shape, dtype=numpy.int64))
        out = numpy.empty(n, numpy.float64)
        filled = 0
        while filled < n:
            want = n - filled
            pairs = max(16, int(0.7 * want) + 8)        # each accepted pair yields two normals
            u = self._symmetric_uniform(pairs)
            v = self._symmetric_uniform(pairs)
            s = u * u + v * v
            keep = (s > 0.0) & (s < 1.0)
            u, v, s = u[keep], v[keep], s[keep]
            f = numpy.sqrt(-2.0 * numpy.log(s) / s)
            z = numpy.stack([u * f, v * f], axis=1).reshape(-1)[:want]
            out[filled:filled + z.size] = z
            filled += z.size
        return out.reshape(shape) * std_dev


def rand_gaussian_torus32_host(rng, message, sigma: float, shape, centered=False):
    """random_numbers.py:134-139"""
    rfloats = rng.gauss(shape, sigma)
    if centered:
        rfloats -= rfloats.mean()
    return (Torus32(message) + double_to_t32(rfloats)).astype(Torus32)


def rand_uniform_bool(thr, rng, shape):
    return thr.to_device(rng.uniform_bool(shape))


def rand_uniform_torus32(thr, rng, shape):
    return thr.to_device(rng.uniform_torus32(shape))


def rand_gaussian_torus32(thr, rng, message, sigma: float, shape, centered=False):
    return thr.to_device(rand_gaussian_torus32_host(rng, message, sigma, shape, centered=centered))
