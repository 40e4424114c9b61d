"""Alias of nufhe_amd.tgsw under the reference's module name (drop-in imports, pickle compatibility)."""
from nufhe_amd.tgsw import *  # noqa: F401,F403
