"""Alias of nufhe_amd.performance under the reference's module name (drop-in imports, pickle compatibility)."""
from nufhe_amd.performance import *  # noqa: F401,F403
