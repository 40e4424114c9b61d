"""Alias of nufhe_amd.gates under the reference's module name (drop-in imports, pickle compatibility)."""
from nufhe_amd.gates import *  # noqa: F401,F403
