"""Alias of nufhe_amd.bootstrap under the reference's module name (drop-in imports, pickle compatibility)."""
from nufhe_amd.bootstrap import *  # noqa: F401,F403
