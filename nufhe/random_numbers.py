"""Alias of nufhe_amd.random_numbers under the reference's module name (drop-in imports, pickle compatibility)."""
from nufhe_amd.random_numbers import *  # noqa: F401,F403
