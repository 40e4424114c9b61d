from nufhe_amd.operators_integer import *  # noqa: F401,F403
from nufhe_amd.operators_integer import uint_min, uintarray_to_bitarray, bitarray_to_uintarray  # noqa: F401
