"""Alias module (nufhe/polynomial_transform.py of the reference): the capability queries its tests import."""
from nufhe_amd.performance import max_supported_transforms_per_block, transform_supported  # noqa: F401
