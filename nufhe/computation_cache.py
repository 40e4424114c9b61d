"""Alias module: the reference's kernel cache does not exist here (kernels are compiled ahead of time)."""
from nufhe_amd.api_high_level import clear_computation_cache  # noqa: F401
