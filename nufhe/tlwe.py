"""Alias of nufhe_amd.tlwe under the reference's module name (drop-in imports, pickle compatibility)."""
from nufhe_amd.tlwe import *  # noqa: F401,F403
