"""Alias of nufhe_amd.polynomials under the reference's module name (drop-in imports)."""
from nufhe_amd.polynomials import *  # noqa: F401,F403
