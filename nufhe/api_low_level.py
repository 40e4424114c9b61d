"""Alias of nufhe_amd.api_low_level under the reference's module name (drop-in imports, pickle compatibility)."""
from nufhe_amd.api_low_level import *  # noqa: F401,F403
