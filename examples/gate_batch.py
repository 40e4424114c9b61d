"""
Independent gates of different kinds as ONE launch: a 1-bit full adder column for a batch of encrypted bits, written
gate by gate and as two `VirtualMachine.gate_batch` calls (no counterpart in the reference's examples: there every gate is
its own chain of launches, nufhe/operators_integer.py:64-95).

    sum   = a XOR b XOR cin         level 1: t = a XOR b, g = a AND b      (independent: one batch)
    carry = (a AND b) OR (t AND cin) level 2: sum = t XOR cin, p = t AND cin (independent: one batch), then carry = g OR p

    python examples/gate_batch.py [--bits 64] [--transform NTT|FFT]
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
    parser.add_argument("--bits", type=int, default=64)
    parser.add_argument("--transform", default="NTT", choices=["NTT", "FFT"])
    opts = parser.parse_args()

    rs = numpy.random.RandomState(1)
    a, b, cin = [rs.randint(0, 2, size=opts.bits).astype(bool) for _ in range(3)]
    context = nufhe.Context()
    secret, cloud = context.make_key_pair(transform_type=opts.transform)
    ea, eb, ec = [context.encrypt(secret, x) for x in (a, b, cin)]
    vm = context.make_virtual_machine(cloud)

    def gate_by_gate():
        t = vm.gate_xor(ea, eb)
        g = vm.gate_and(ea, eb)
        s = vm.gate_xor(t, ec)
        p = vm.gate_and(t, ec)
        return s, vm.gate_or(g, p)

    def batched():
        t, g = vm.gate_batch([('gate_xor', ea, eb), ('gate_and', ea, eb)])
        s, p = vm.gate_batch([('gate_xor', t, ec), ('gate_and', t, ec)])
        return s, vm.gate_or(g, p)

    results = {}
    for name, fn in (("gate by gate", gate_by_gate), ("two batches + one gate", batched)):
        fn()                                             # warm-up
        context.thread.synchronize()
        started = time.perf_counter()
        results[name] = fn()
        context.thread.synchronize()
        print("%-24s %.2f ms" % (name, 1e3 * (time.perf_counter() - started)))
    for x, y in zip(results["gate by gate"], results["two batches + one gate"]):
        assert x == y                                    # the same ciphertext words
    s, c = results["two batches + one gate"]
    assert (context.decrypt(secret, s) == (a ^ b ^ cin)).all()
    assert (context.decrypt(secret, c) == ((a & b) | ((a ^ b) & cin))).all()
    print("full adder on %d encrypted bit columns OK (%s)" % (opts.bits, opts.transform))


if __name__ == "__main__":
    main()
