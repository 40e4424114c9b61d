"""
PerformanceParameters of the reference (nufhe/performance.py:22-236).  Its code-generation knobs -- constant
memory, transforms per block, PTX vs C arithmetic -- select between CUDA/OpenCL variants that do not exist in the
ahead-of-time compiled gfx950 library: they are accepted and ignored.  ``single_kernel_bootstrap`` is honoured:
False runs a gate as the reference's multi-kernel sequence, one launch per step (nufhe_amd/bootstrap.py,
nufhe_amd/gates.py), True / None takes the fused kernels.
"""


class PerformanceParametersForDevice:

    def __init__(self, nufhe_params, **kwds):
        self.nufhe_params = nufhe_params
        self.single_kernel_bootstrap = True
        self.__dict__.update(kwds)


class PerformanceParameters:

    def __init__(
            self, nufhe_params,
            ntt_base_method=None, ntt_mul_method=None, ntt_lsh_method=None,
            use_constant_memory_multi_iter=None, use_constant_memory_single_iter=None,
            transforms_per_block=None, single_kernel_bootstrap=None, low_end_device=None):
        self.nufhe_params = nufhe_params
        self._kwds = dict(
            ntt_base_method=ntt_base_method, ntt_mul_method=ntt_mul_method,
            ntt_lsh_method=ntt_lsh_method,
            use_constant_memory_multi_iter=use_constant_memory_multi_iter,
            use_constant_memory_single_iter=use_constant_memory_single_iter,
            transforms_per_block=transforms_per_block, low_end_device=low_end_device)
        self.single_kernel_bootstrap = single_kernel_bootstrap

    def for_device(self, device_params=None):
        res = PerformanceParametersForDevice(self.nufhe_params, **self._kwds)
        if self.single_kernel_bootstrap is not None:
            res.single_kernel_bootstrap = bool(self.single_kernel_bootstrap)
        return res
