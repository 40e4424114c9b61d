"""
Client / cloud round trip through serialized keys and ciphertexts (counterpart of the reference's
`examples/serialization.py`).  The byte streams use the reference's pickle record layout
(`nufhe_amd/serialization.py`), so the files written here load in the reference and vice versa.

  client: make keys, encrypt, write secret_key / cloud_key / ciphertext files
  cloud : load cloud key + ciphertexts, evaluate NAND, write the result
  client: load secret key + result, decrypt, compare

    python examples/serialization.py [--bits 32] [--dir /tmp/nufhe_demo]
"""
import argparse
import os
import sys
import tempfile

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
import nufhe


def client_prepare(folder, nbits):
    context = nufhe.Context()
    secret, cloud = context.make_key_pair()
    plain = numpy.random.randint(0, 2, size=(2, nbits)).astype(bool)
    pieces = {"secret_key": secret, "cloud_key": cloud,
              "ciphertext1": context.encrypt(secret, plain[0]),
              "ciphertext2": context.encrypt(secret, plain[1])}
    for name, obj in pieces.items():
        with open(os.path.join(folder, name), "wb") as stream:
            obj.dump(stream)
    return ~(plain[0] & plain[1])


def cloud_process(folder):
    context = nufhe.Context()
    with open(os.path.join(folder, "cloud_key"), "rb") as stream:
        cloud = context.load_cloud_key(stream)
    machine = context.make_virtual_machine(cloud)
    inputs = []
    for name in ("ciphertext1", "ciphertext2"):
        with open(os.path.join(folder, name), "rb") as stream:
            inputs.append(machine.load_ciphertext(stream))
    with open(os.path.join(folder, "result"), "wb") as stream:
        machine.gate_nand(*inputs).dump(stream)


def client_verify(folder, expected):
    context = nufhe.Context()
    with open(os.path.join(folder, "secret_key"), "rb") as stream:
        secret = context.load_secret_key(stream)
    with open(os.path.join(folder, "result"), "rb") as stream:
        result = context.load_ciphertext(stream)
    if not numpy.array_equal(context.decrypt(secret, result), expected):
        raise SystemExit("decrypted result differs from NAND of the plaintexts")


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--bits", type=int, default=32)
    parser.add_argument("--dir", default=None)
    opts = parser.parse_args()
    folder = opts.dir or tempfile.mkdtemp(prefix="nufhe_demo_")
    os.makedirs(folder, exist_ok=True)
    expected = client_prepare(folder, opts.bits)
    cloud_process(folder)
    client_verify(folder, expected)
    sizes = {name: os.path.getsize(os.path.join(folder, name)) for name in sorted(os.listdir(folder))}
    print("serialization round trip OK;", ", ".join("%s %.1f MB" % (k, v / 1e6) for k, v in sizes.items()))
    if opts.dir is None:
        for name in sizes:
            os.remove(os.path.join(folder, name))
        os.rmdir(folder)


if __name__ == "__main__":
    main()
