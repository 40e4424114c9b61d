"""Alias of nufhe_amd.numeric_functions under the reference's module name (drop-in imports, pickle compatibility)."""
from nufhe_amd.numeric_functions import *  # noqa: F401,F403
