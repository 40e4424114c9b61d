"""Alias module (nufhe/blind_rotate.py of the reference): the capability query its tests import."""
from nufhe_amd.performance import single_kernel_bootstrap_supported  # noqa: F401
