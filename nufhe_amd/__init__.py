"""
nufhe_amd -- MI355X (gfx950) implementation of nufhe's bootstrapped-gate hot path behind the
reference's Python API (reference: nufhe/__init__.py).  ``import nufhe_amd as nufhe`` is a
drop-in for the gate path: Context / VirtualMachine / gate_* / encrypt / decrypt / keys.
"""

from .random_numbers import DeterministicRNG, SecureRNG
from .api_low_level import (
    NuFHEParameters,
    NuFHESecretKey,
    NuFHECloudKey,
    make_key_pair,
    encrypt,
    decrypt,
    empty_ciphertext,
    )
from .api_high_level import (
    Context,
    VirtualMachine,
    DeviceID,
    find_devices,
    clear_computation_cache,
    )
from .lwe import LweSampleArray, concatenate
from .gates import (
    gate_nand, gate_or, gate_and, gate_xor, gate_xnor, gate_not, gate_copy, gate_constant,
    gate_nor, gate_andny, gate_andyn, gate_orny, gate_oryn, gate_mux,
    )
from .performance import PerformanceParameters
from .graph import GateGraph

__version__ = '0.4.0'
