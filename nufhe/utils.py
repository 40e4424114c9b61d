"""Alias of nufhe_amd.utils under the reference's module name (drop-in imports, pickle compatibility)."""
from nufhe_amd.utils import *  # noqa: F401,F403
