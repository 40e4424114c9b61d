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


# The reference's capability queries (nufhe/blind_rotate.py:192-203, nufhe/polynomial_transform.py:23-39): its tests
# skip configurations a device cannot run.  The gfx950 library is compiled ahead of time for one device class and
# serves every configuration the reference's gate tests ask for.

def single_kernel_bootstrap_supported(nufhe_params, device_params):
    """The fused bootstrap kernels exist for both transforms and tlwe_mask_size 1 and 2 (csrc/kernels.hip)."""
    return True


def transform_supported(device_params, transform_type):
    return transform_type in ('NTT', 'FFT')


def max_supported_transforms_per_block(device_params, transform_type):
    """The reference sizes its work-groups by this (performance.py:159-180); here a work-group holds 8 one-wave
    transforms (one per ciphertext bit) whatever the transform -- the knob it feeds, `transforms_per_block`, is ignored."""
    return 8
