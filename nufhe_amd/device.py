"""
DeviceThread: the object passed wherever the reference takes a Reikna ``Thread`` (``thr``).
One GPU + one in-order HIP stream (reference: reikna.cluda Thread, used all over
nufhe/api_high_level.py:130-181).  Device memory is held in torch tensors (plumbing only: an
allocator and a stream); every computation of the hot path goes to libnufhe_hip.so through the
C ABI with raw pointers.
"""

import ctypes
import weakref

import numpy
import torch

from . import _lib
from ._lib import NufheLwe


class DeviceParams:
    """Stand-in for Reikna's device_params (only consumed by PerformanceParameters.for_device)."""

    def __init__(self, name, compute_units):
        self.name = name
        self.compute_units = compute_units
        self.max_work_group_size = 1024
        self.local_mem_size = 160 * 1024


class _Api:
    """``thread.api.get_id()`` of a Reikna thread (the reference's tests compare it with ``cuda_id()`` to skip their
    PTX-only variants): this backend is neither CUDA nor OpenCL."""

    @staticmethod
    def get_id():
        return 'hip'


_NP_TO_TORCH = {
    numpy.dtype('int32'): torch.int32,
    numpy.dtype('float32'): torch.float32,
    numpy.dtype('float64'): torch.float64,
    numpy.dtype('complex128'): torch.complex128,
    numpy.dtype('uint64'): torch.uint64 if hasattr(torch, 'uint64') else torch.int64,
    numpy.dtype('int64'): torch.int64,
    numpy.dtype('bool'): torch.bool,
}


class DeviceThread:

    def __init__(self, device_index=0):
        lib = _lib.lib()   # raises if the HIP library is missing: no CPU fallback
        count = ctypes.c_int(0)
        _lib.check(lib.nufhe_device_count(ctypes.byref(count)))
        if count.value == 0 or not torch.cuda.is_available():
            raise _lib.NufheError("no MI355X / HIP device available (nufhe_amd has no CPU fallback)")
        self.device_index = int(device_index)
        self.device = torch.device('cuda', self.device_index)
        torch.cuda.set_device(self.device)
        self._torch_stream = torch.cuda.current_stream(self.device)
        handle = ctypes.c_void_p()
        # kernels are enqueued on torch's current stream (handle 0 = the default stream), so torch's
        # copies/allocations and the library's launches are ordered without extra synchronisation
        _lib.check(lib.nufhe_ctx_create(
            self.device_index, ctypes.c_void_p(self._torch_stream.cuda_stream), 0, ctypes.byref(handle)))
        self.handle = handle
        ref = weakref.ref(self)
        _lib.register_stream_guard(handle, lambda: (ref() is not None) and ref().check_stream())
        props = torch.cuda.get_device_properties(self.device)
        self.device_params = DeviceParams(props.name, props.multi_processor_count)
        self.api = _Api()
        self._released = False
        self._cloud_keys = weakref.WeakSet()     # native cloud keys living in this context (freed by release())

    def check_stream(self):
        """The library enqueues on the torch stream that was current when this object was created; torch's
        own copies / allocations must run on the same stream to stay ordered with the kernels (and for the
        caching allocator not to recycle a temporary a kernel still reads).  Raises if the caller switched
        streams, e.g. inside ``with torch.cuda.stream(s):``."""
        if torch.cuda.current_stream(self.device) != self._torch_stream:
            raise RuntimeError(
                "nufhe_amd: the current torch stream differs from the one this DeviceThread was created on; "
                "create the DeviceThread (Context) inside the torch.cuda.stream(...) block that uses it")

    # ---- memory -------------------------------------------------------------------------
    def array(self, shape, dtype):
        return torch.empty(tuple(shape), dtype=_NP_TO_TORCH[numpy.dtype(dtype)], device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(tuple(shape), dtype=_NP_TO_TORCH[numpy.dtype(dtype)], device=self.device)

    def empty_like(self, arr):
        return torch.empty_like(arr)

    def to_device(self, arr):
        arr = numpy.ascontiguousarray(arr)
        if arr.dtype == numpy.uint64 and not hasattr(torch, 'uint64'):
            arr = arr.view(numpy.int64)
        return torch.from_numpy(arr).to(self.device)

    def from_device(self, arr):
        return arr.detach().cpu().numpy()

    def synchronize(self):
        _lib.call("nufhe_ctx_synchronize", self.handle)

    def tuning(self):
        """The batch-size switch points of this context as a dict (``nufhe_ctx_get_tuning``): derived from the device
        (architecture name + CU count) when the context was created; ``measured`` says whether that part has an entry
        in the library's table of measured switch points."""
        import ctypes
        t = _lib.NufheTuning()
        _lib.check(_lib.lib().nufhe_ctx_get_tuning(self.handle, ctypes.byref(t)))
        out = {name: getattr(t, name) for name, _ in t._fields_}
        out['arch_name'] = out['arch_name'].decode()
        return out

    def release(self):
        if not self._released and self.handle:
            for key in list(self._cloud_keys):    # ~100 MB of device memory each: free them with the context
                key.destroy()
            _lib.unregister_stream_guard(self.handle)
            _lib.lib().nufhe_ctx_destroy(self.handle)
            self._released = True

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def ptr(tensor):
    return ctypes.c_void_p(tensor.data_ptr() if tensor is not None else 0)


def int32_operand(name, tensor, device, shape=None):
    """A key / message tensor about to be handed to a kernel that reads int32.  Other integer / bool tensors are
    CONVERTED (the reference's ``astype(Torus32)``, lwe.py:265-270, tgsw.py:155-161) -- handing their bytes over
    as they are would be reinterpreted silently --, floating-point ones are refused; returns the tensor contiguous
    on this device.  ``shape``: expected shape."""
    if not isinstance(tensor, torch.Tensor):
        raise TypeError("%s: expected a torch tensor, got %s" % (name, type(tensor).__name__))
    if tensor.dtype != torch.int32:
        if tensor.dtype.is_floating_point or tensor.dtype.is_complex:
            raise TypeError("%s: expected an integer tensor, got %s" % (name, tensor.dtype))
        tensor = tensor.to(torch.int32)
    if tensor.device != device:
        raise ValueError("%s lives on %s, the context on %s" % (name, tensor.device, device))
    if shape is not None and tuple(tensor.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(tensor.shape), tuple(shape)))
    return tensor.contiguous()


def lwe_desc(a, b, cv, size):
    """Builds the C descriptor of an LWE sample batch from (a [B, size], b [B], cv [B]) device
    tensors that are 2D/1D views with a contiguous last axis (stride-0 batch axes broadcast)."""
    assert a.dim() == 2 and b.dim() == 1
    if a.shape[1] != size or (a.shape[0] > 0 and a.shape[1] > 1 and a.stride(1) != 1):     # (an empty batch has no layout)
        raise ValueError("LWE mask array must have a contiguous last axis of length %d" % size)
    if cv is not None:
        assert cv.dim() == 1 and (cv.stride(0) == b.stride(0) or cv.shape[0] <= 1)
    return NufheLwe(
        a=a.data_ptr(), b=b.data_ptr(), cv=(cv.data_ptr() if cv is not None else None),
        a_stride=a.stride(0) if a.shape[0] > 1 else (a.stride(0) if a.shape[0] == 1 else 0),
        b_stride=b.stride(0) if b.shape[0] > 1 else (b.stride(0) if b.shape[0] == 1 else 0),
        size=size)
