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


def test_gather_threads_single_process_collection():
    """multi_gpu.gather_threads / nufhe_gather: the reference's own multi-GPU shape -- one process, one Thread object per
    GPU, the main thread collecting the result slices (examples/multi_gpu.py:46-114) -- with device-to-device copies
    instead of pickles.  On this one-GPU box the three "GPUs" are three contexts of device 0, two of them on streams of
    their own, so the copy-on-the-source-stream + destination-waits-for-event path runs for real; an even and a ragged
    split, a strided slice (packed through a temporary) and an empty slice are covered, and the gathered ciphertext equals
    the unsharded gate word for word."""
    import torch
    import nufhe_amd as nufhe
    from nufhe_amd import multi_gpu
    from nufhe_amd.device import DeviceThread
    main = DeviceThread(0)
    ctx = nufhe.Context(rng=nufhe.DeterministicRNG(5), thread=main)
    sk, ck = ctx.make_key_pair()
    blob = ck.dumps()
    rs = numpy.random.RandomState(9)
    B = 70
    m1 = rs.randint(0, 2, B).astype(bool); m2 = rs.randint(0, 2, B).astype(bool)
    c1 = ctx.encrypt(sk, m1); c2 = ctx.encrypt(sk, m2)
    whole = ctx.make_virtual_machine(ck).gate_nand(c1, c2)
    streams = [None, torch.cuda.Stream(), torch.cuda.Stream()]
    bounds = [(0, 30), (30, 30), (30, 70)]                     # even-ish, EMPTY, ragged
    parts = []
    for s, (lo, hi) in zip(streams, bounds):
        scope = torch.cuda.stream(s) if s is not None else torch.cuda.stream(torch.cuda.current_stream())
        with scope:
            thr = DeviceThread(0) if s is not None else main
            wctx = nufhe.Context(rng=nufhe.DeterministicRNG(1), thread=thr)
            wck = ck if s is None else wctx.load_cloud_key(blob)
            vm = wctx.make_virtual_machine(wck)
            if s is not None:
                s.wait_stream(torch.cuda.default_stream())         # the inputs were made on the main stream
            parts.append((thr, vm.gate_nand(c1[lo:hi], c2[lo:hi]), s))
    # one worker's slice handed over as a strided view (every second bit of a double-size result)
    with torch.cuda.stream(streams[2]):
        thr2, r2, _ = parts[2]
        wide = nufhe.api_low_level.empty_ciphertext(thr2, ck.params, (2 * 40,))
        wide[::2] = r2
        parts[2] = (thr2, wide[::2], streams[2])
    # the collection itself is called under the DESTINATION's stream (main)
    full = multi_gpu.gather_threads(main, [(t, r) for t, r, _ in parts])
    assert full.shape == (B,)
    assert full == whole
    assert (ctx.decrypt(sk, full) == ~(m1 & m2)).all()
    with pytest.raises(ValueError):
        multi_gpu.gather_threads(main, [])
