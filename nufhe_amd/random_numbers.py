"""
Host random number generators (reference: nufhe/random_numbers.py:46-151).  As in the reference,
randomness is drawn on the host and uploaded; the draw ORDER of key generation and encryption
 is preserved so that one seed gives the same keys and ciphertexts.
"""

import random
from os import urandom

import numpy

from .numeric_functions import Torus32, Int32, double_to_t32

Float = numpy.dtype('float64')
MantissaInt = numpy.dtype('uint64')
BPF = numpy.finfo(Float).nmant + 1
RECIP_BPF = 2**(-BPF)


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
    """A cryptographically secure RNG backed by the OS (random_numbers.py:65-130)."""

    def __init__(self):
        self.rng = random.SystemRandom()

    def uniform_bool(self, shape):
        length = int(numpy.prod(shape))
        nbytes = (length - 1) // 8 + 1
        bits = numpy.unpackbits(numpy.frombuffer(urandom(nbytes), numpy.uint8))[:length]
        return bits.reshape(shape).astype(Int32)

    def uniform_torus32(self, shape):
        length = int(numpy.prod(shape))
        return numpy.frombuffer(urandom(length * 4), Int32).reshape(shape).copy()

    def _uniform_float(self, length):
        # open interval (0, 1): drop one extra bit, then shift by half a step
        mantissa = numpy.frombuffer(urandom(length * MantissaInt.itemsize), MantissaInt)
        mantissa = mantissa >> numpy.uint64(MantissaInt.itemsize * 8 - (BPF - 1))
        mantissa = mantissa * numpy.uint64(2) + numpy.uint64(1)
        return mantissa * RECIP_BPF

    def gauss(self, shape, std_dev):
        orig_length = int(numpy.prod(shape))
        length = orig_length + orig_length % 2
        u1 = self._uniform_float(length // 2)
        u2 = self._uniform_float(length // 2)
        r = (-2 * numpy.log(u1))**0.5
        theta = 2 * numpy.pi * u2
        result = numpy.concatenate([r * numpy.cos(theta), r * numpy.sin(theta)])[:orig_length]
        return result.reshape(shape) * std_dev


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
