"""
High-level API: Context / VirtualMachine / device discovery
(reference: nufhe/api_high_level.py:31-363).
"""

import ctypes

import numpy

from . import _lib
from . import gates
from .api_low_level import (
    NuFHEParameters, NuFHESecretKey, NuFHECloudKey, encrypt, decrypt, empty_ciphertext)
from .device import DeviceThread
from .gates import get_shape, result_shape
from .lwe import LweSampleArray
from .performance import PerformanceParameters
from .random_numbers import DeterministicRNG


class DeviceID:
    """
    Identifier of a computation device (nufhe/api_high_level.py:85-127); picklable, so it can be
    passed to another thread or process and used to create a :class:`Context`.
    """

    def __init__(self, device_index, device_name):
        self.api_id = 'HIP'
        self.platform_id = 0
        self.device_id = int(device_index)
        self.api_name = 'HIP'
        self.platform_name = 'ROCm'
        self.device_name = device_name

    def __str__(self):
        return "DeviceID({api}, {platform}, {device})".format(
            api=self.api_name, platform=self.platform_name, device=self.device_name)


def find_devices(api=None, include_devices=None, exclude_devices=None,
                 include_platforms=None, exclude_platforms=None):
    """
    Returns the list of usable GPUs as :class:`DeviceID` objects (nufhe/api_high_level.py:45-82).
    ``api`` may be ``None`` or ``'HIP'``; the name masks filter on the device name.
    """
    if api not in (None, 'HIP'):
        raise ValueError("Uknonwn GPGPU API identifier: " + str(api))
    lib = _lib.lib()
    count = ctypes.c_int(0)
    _lib.check(lib.nufhe_device_count(ctypes.byref(count)))
    devices = []
    for idx in range(count.value):
        buf = ctypes.create_string_buffer(256)
        _lib.check(lib.nufhe_device_name(idx, buf, 256))
        name = buf.value.decode()
        if include_devices is not None and not any(mask in name for mask in include_devices):
            continue
        if exclude_devices is not None and any(mask in name for mask in exclude_devices):
            continue
        devices.append(DeviceID(idx, name))
    if len(devices) == 0:
        raise ValueError("No devices found satisfying the given search criteria")
    return devices


def clear_computation_cache(thread=None):
    """No-op: kernels are compiled ahead of time (reference: nufhe/computation_cache.py:32-44)."""


class Context:
    """
    An execution environment on one GPU (nufhe/api_high_level.py:130-299).

    :param rng: a random number generator (``DeterministicRNG`` by default).
    :param thread: an existing :class:`nufhe_amd.device.DeviceThread`.
    :param device_id: a :class:`DeviceID` from :func:`find_devices`.
    """

    def __init__(self, rng=None, thread=None, device_id: DeviceID=None, api=None, interactive=False,
                 include_devices=None, exclude_devices=None,
                 include_platforms=None, exclude_platforms=None):
        if rng is None:
            rng = DeterministicRNG()
        if thread is not None:
            pass
        elif device_id is not None:
            thread = DeviceThread(device_id.device_id)
        else:
            devices = find_devices(
                api=api, include_devices=include_devices, exclude_devices=exclude_devices,
                include_platforms=include_platforms, exclude_platforms=exclude_platforms)
            thread = DeviceThread(devices[0].device_id)
        self.rng = rng
        self.thread = thread

    def make_secret_key(self, **params):
        nufhe_params = NuFHEParameters(**params)
        return NuFHESecretKey.from_rng(self.thread, nufhe_params, self.rng)

    def make_cloud_key(self, secret_key: NuFHESecretKey):
        return NuFHECloudKey.from_rng(self.thread, secret_key.params, self.rng, secret_key)

    def make_key_pair(self, **params):
        secret_key = self.make_secret_key(**params)
        cloud_key = self.make_cloud_key(secret_key)
        return secret_key, cloud_key

    def encrypt(self, secret_key: NuFHESecretKey, message):
        return encrypt(self.thread, self.rng, secret_key, message)

    def decrypt(self, secret_key: NuFHESecretKey, ciphertext: LweSampleArray):
        return decrypt(self.thread, secret_key, ciphertext)

    def make_virtual_machine(self, cloud_key: NuFHECloudKey, perf_params: PerformanceParameters=None):
        return VirtualMachine(self.thread, cloud_key, perf_params=perf_params)

    def load_ciphertext(self, file_or_bytestring):
        if isinstance(file_or_bytestring, bytes):
            return LweSampleArray.loads(file_or_bytestring, self.thread)
        return LweSampleArray.load(file_or_bytestring, self.thread)

    def load_secret_key(self, file_or_bytestring):
        if isinstance(file_or_bytestring, bytes):
            return NuFHESecretKey.loads(file_or_bytestring, self.thread)
        return NuFHESecretKey.load(file_or_bytestring, self.thread)

    def load_cloud_key(self, file_or_bytestring):
        if isinstance(file_or_bytestring, bytes):
            return NuFHECloudKey.loads(file_or_bytestring, self.thread)
        return NuFHECloudKey.load(file_or_bytestring, self.thread)


class VirtualMachine:
    """
    Executes gates on ciphertexts with an encapsulated cloud key
    (nufhe/api_high_level.py:302-363).

    .. method:: gate_<operator>(*args, dest: LweSampleArray=None)
    """

    def __init__(self, thread, cloud_key: NuFHECloudKey, perf_params: PerformanceParameters=None):
        if perf_params is None:
            perf_params = PerformanceParameters(cloud_key.params)
        perf_params = perf_params.for_device(thread.device_params)
        self.thread = thread
        self.params = cloud_key.params
        self.cloud_key = cloud_key
        self.perf_params = perf_params

    def empty_ciphertext(self, shape):
        return empty_ciphertext(self.thread, self.params, shape)

    def load_ciphertext(self, file):
        return LweSampleArray.load(file, self.thread)

    def _gate(self, name, *args, dest: LweSampleArray=None):
        if dest is None:
            shapes = [get_shape(arg) for arg in args]
            dest = self.empty_ciphertext(result_shape(*shapes))
        gate_func = getattr(gates, name)
        gate_func(self.thread, self.cloud_key, dest, *args, perf_params=self.perf_params)
        return dest

    def gate_batch(self, jobs):
        """
        Runs a list of INDEPENDENT gates as one bootstrap launch (:func:`nufhe_amd.gates.gate_batch`):
        ``jobs`` = [('gate_nand', a, b), ('gate_mux', s, x, y), ('gate_xor', a, b, dest), ...] -- the gate's name, its
        ciphertext arguments and, optionally, a destination as the last element.  Returns the list of results, in
        order.  Small independent gates (up to about one bit per compute unit in total) finish in the time of one.
        """
        prepared, results = [], []
        for job in jobs:
            name, args = job[0], list(job[1:])
            arity = 3 if name == 'gate_mux' else 2
            if len(args) not in (arity, arity + 1):
                raise ValueError("%s takes %d ciphertexts (+ an optional destination)" % (name, arity))
            dest = args.pop() if len(args) == arity + 1 else None
            if dest is None:
                dest = self.empty_ciphertext(result_shape(*[get_shape(arg) for arg in args]))
            prepared.append((name, dest) + tuple(args))
            results.append(dest)
        gates.gate_batch(self.thread, self.cloud_key, prepared, perf_params=self.perf_params)
        return results

    def __getattr__(self, name):
        if name.startswith('gate_'):
            return lambda *args, **kwds: self._gate(name, *args, **kwds)
        raise AttributeError(name)
