"""
The same NAND through the low-level functions: the caller owns the device thread, the RNG and the
destination ciphertext (counterpart of the reference's `examples/gate_nand_low_level.py`; there the
thread is a Reikna `Thread`, here it is a `nufhe_amd.device.DeviceThread` on one MI355X).

    python examples/gate_nand_low_level.py [--bits 32]
"""
import argparse
import os
import sys

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
import nufhe
from nufhe_amd.device import DeviceThread


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--bits", type=int, default=32)
    opts = parser.parse_args()

    plain = numpy.random.randint(0, 2, size=(2, opts.bits)).astype(bool)
    expected = ~(plain[0] & plain[1])

    device = DeviceThread(0)
    rng = nufhe.DeterministicRNG()
    secret, cloud = nufhe.make_key_pair(device, rng)
    enc = [nufhe.encrypt(device, rng, secret, row) for row in plain]

    out = nufhe.empty_ciphertext(device, cloud.params, enc[0].shape)
    nufhe.gate_nand(device, cloud, out, enc[0], enc[1])

    if not numpy.array_equal(nufhe.decrypt(device, secret, out), expected):
        raise SystemExit("NAND mismatch")
    print("low-level gate_nand OK on %d bits" % opts.bits)


if __name__ == "__main__":
    main()
