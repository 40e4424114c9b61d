"""
Drop-in contract under threads (SURVEY §8b "Threading", reference examples/multi_gpu.py:46-114): one
`nufhe.Context(device_id=...)` per Python thread, the cloud key and the ciphertext halves handed over as
pickled bytes, the results handed back as bytes, the main thread decrypting.  On a 1-GPU box both
workers use devices[0] (the reference asserts two devices; the contract under test is the
context-per-thread API, not the second device).  Goes through the `nufhe` alias package, i.e. exactly
the names the reference's example imports.
"""
from queue import Queue
from threading import Thread

import numpy
import pytest

pytestmark = pytest.mark.gpu


class MyThread:
    """receives the value returned by the worker (same shape as the helper in the reference's example)"""

    def __init__(self, target, args=()):
        self.return_queue = Queue()
        self.thread = Thread(target=self._run, args=(target,) + tuple(args))

    def _run(self, target, *args):
        try:
            self.return_queue.put(target(*args))
        except BaseException as e:       # surface worker failures in the main thread
            self.return_queue.put(e)

    def start(self):
        self.thread.start()
        return self

    def join(self):
        ret = self.return_queue.get()
        self.thread.join()
        if isinstance(ret, BaseException):
            raise ret
        return ret


def worker(device_id, cloud_key_cpu, ciphertext1_cpu, ciphertext2_cpu):
    import nufhe
    ctx = nufhe.Context(device_id=device_id)
    cloud_key = ctx.load_cloud_key(cloud_key_cpu)
    ciphertext1 = ctx.load_ciphertext(ciphertext1_cpu)
    ciphertext2 = ctx.load_ciphertext(ciphertext2_cpu)
    vm = ctx.make_virtual_machine(cloud_key)
    result = vm.gate_nand(ciphertext1, ciphertext2)
    return result.dumps()


def test_two_contexts_in_two_threads():
    import nufhe
    size = 64
    rs = numpy.random.RandomState(12)
    bits1 = rs.randint(0, 2, size=size).astype(bool).tolist()
    bits2 = rs.randint(0, 2, size=size).astype(bool).tolist()
    reference = [not (b1 and b2) for b1, b2 in zip(bits1, bits2)]

    ctx = nufhe.Context()
    secret_key, cloud_key = ctx.make_key_pair()
    ciphertext1 = ctx.encrypt(secret_key, bits1)
    ciphertext2 = ctx.encrypt(secret_key, bits2)
    ck = cloud_key.dumps()
    ct1_part1 = ciphertext1[:size // 2].dumps(); ct1_part2 = ciphertext1[size // 2:].dumps()
    ct2_part1 = ciphertext2[:size // 2].dumps(); ct2_part2 = ciphertext2[size // 2:].dumps()

    devices = nufhe.find_devices()
    dev = [devices[0], devices[1] if len(devices) >= 2 else devices[0]]
    t1 = MyThread(target=worker, args=(dev[0], ck, ct1_part1, ct2_part1)).start()
    t2 = MyThread(target=worker, args=(dev[1], ck, ct1_part2, ct2_part2)).start()
    result_part1 = ctx.load_ciphertext(t1.join())
    result_part2 = ctx.load_ciphertext(t2.join())
    r1 = ctx.decrypt(secret_key, result_part1)
    r2 = ctx.decrypt(secret_key, result_part2)
    assert r1.tolist() + r2.tolist() == reference
    # same bits through one context: identical ciphertexts (the keys the workers loaded are the same key)
    whole = ctx.make_virtual_machine(cloud_key).gate_nand(ciphertext1, ciphertext2)
    assert whole[:size // 2] == result_part1 and whole[size // 2:] == result_part2
    # a DeviceID survives pickling (it is what the reference ships to worker processes)
    import pickle
    assert str(pickle.loads(pickle.dumps(devices[0]))) == str(devices[0])
