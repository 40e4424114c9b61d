"""Alias of nufhe_amd.lwe under the reference's module name (drop-in imports, pickle compatibility)."""
from nufhe_amd.lwe import *  # noqa: F401,F403
