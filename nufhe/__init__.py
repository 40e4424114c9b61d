"""
`import nufhe` drop-in: the reference's package name, served by nufhe_amd (MI355X / gfx950).
Same public names as the reference's nufhe/__init__.py.  The sub-modules are aliases as well, so
that pickles written by the reference (which name classes such as ``nufhe.lwe.LweParams``) load
here and pickles written here load in the reference.
"""
from nufhe_amd import *  # noqa: F401,F403
from nufhe_amd import __version__  # noqa: F401
