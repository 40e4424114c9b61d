"""
Bootstrapped NAND on a batch of encrypted bits through the nufhe API surface
(what the reference's `examples/gate_nand.py` demonstrates; BASELINE config 1/2 shape).
`import nufhe` resolves to the alias package of this repository, served by nufhe_amd on MI355X.

    python examples/gate_nand.py [--bits 4096] [--transform NTT|FFT]
"""
import argparse
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
import nufhe


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--bits", type=int, default=32)
    parser.add_argument("--transform", default="NTT", choices=["NTT", "FFT"])
    opts = parser.parse_args()

    plain_a = numpy.random.randint(0, 2, size=opts.bits).astype(bool)
    plain_b = numpy.random.randint(0, 2, size=opts.bits).astype(bool)

    context = nufhe.Context()
    secret, cloud = context.make_key_pair(transform_type=opts.transform)
    enc_a = context.encrypt(secret, plain_a)
    enc_b = context.encrypt(secret, plain_b)

    machine = context.make_virtual_machine(cloud)
    machine.gate_nand(enc_a, enc_b)                      # warm-up (first launch)
    context.thread.synchronize()
    started = time.perf_counter()
    enc_out = machine.gate_nand(enc_a, enc_b)
    context.thread.synchronize()
    elapsed = time.perf_counter() - started

    plain_out = context.decrypt(secret, enc_out)
    if not numpy.array_equal(plain_out, ~(plain_a & plain_b)):
        raise SystemExit("NAND mismatch")
    print("gate_nand OK on %d bits (%s): %.3f ms, %.4f ms/bit" % (
        opts.bits, opts.transform, 1e3 * elapsed, 1e3 * elapsed / opts.bits))


if __name__ == "__main__":
    main()
