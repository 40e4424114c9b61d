"""
Multi-GPU data parallelism over independent ciphertext bits (reference: examples/multi_gpu.py:46-114,
which ships pickled slices between Python threads through host memory).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU
tests).  The batch is cut into contiguous per-rank slices, the cloud key is replicated, every rank
runs its gates locally with no data-path collective, and the only communication is the result
gather to one rank (RCCL gather = grouped send/recv over xGMI).

Two forms of that gather:
  * ``gather_arrays`` / ``gather_ciphertext``: one blocking collective per array (a, b, variances);
  * ``PackedCiphertext`` + ``gather_packed_async``: the three arrays of a result are views of ONE
    int32 buffer (2008 bytes per bit), so a step is one collective, started with ``async_op=True``:
    RCCL runs it on its own stream after the gate that produced the buffer and the next gate
    overlaps it.  Two such buffers alternate (``bench.py``); ``AsyncGather.wait`` orders the
    caller's stream behind the collective before a buffer is reused.
The functions contain no device-specific code: they work on whatever tensors they are given.
"""

import torch


def launch_ranks(script: str, argv, nproc: int, backend: str = "nccl"):
    """
    Starts ``nproc`` ranks of ``script`` on THIS node (one process per GPU) and returns the exit status:
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node nproc --rdzv-backend=c10d
    --rdzv-endpoint=127.0.0.1:0 --local-addr 127.0.0.1 script argv...`` (the store picks its own free port).  This is what lets ``python bench.py --gpus N`` and
    ``python examples/multi_gpu.py --gpus N`` be complete commands, like the reference's example that
    starts its own per-GPU workers (examples/multi_gpu.py:86-114).  A process that already has ``RANK``
    in its environment must not call this (it IS a rank).

    With the ``nccl`` (= RCCL) backend every rank needs its own GPU: fewer visible devices than ranks is
    an error here, before anything is started.  ``gloo`` is the test route (ranks may share a GPU).
    """
    import os
    import subprocess
    import sys
    if "RANK" in os.environ:
        raise RuntimeError("launch_ranks called from inside a rank (RANK is set)")
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    if backend == "nccl" and torch.cuda.device_count() < nproc:
        raise RuntimeError("%d ranks over RCCL need %d GPUs, this node shows %d (set NUFHE_BENCH_BACKEND=gloo "
                           "for the shared-GPU test route)" % (nproc, nproc, torch.cuda.device_count()))
    # the rendezvous store binds port 0 itself (c10d backend): no window between "find a free port" and "use it" in which
    # a concurrent launch on this node could take it; 127.0.0.1 throughout (the container's hostname may not resolve)
    import uuid
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--rdzv-backend=c10d", "--rdzv-endpoint=127.0.0.1:0", "--rdzv-id", uuid.uuid4().hex,
           "--local-addr", "127.0.0.1", script] + list(argv)
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // nproc)))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between the ranks of one node
    return subprocess.call(cmd, env=env)


def broadcast_cloud_key(thr, cloud_key, src: int = 0, group=None):
    """
    Replicates a cloud key from rank ``src`` to every rank of the group as ONE device broadcast of the key image
    (98.6 MB for the default parameters: bootstrapping key in the kernels' layout + keyswitch key; RCCL over xGMI
    with the ``nccl`` backend) -- SURVEY 8e's "ncclBroadcast from rank 0".  The reference pickles the key through
    the host for every worker (examples/multi_gpu.py:86-107); that route stays available as dumps / load_cloud_key.

    ``cloud_key`` is the key on ``src`` and ignored (may be None) elsewhere; returns the rank's key.  The few
    parameter bytes travel as a pickled object first so that the receivers can size their buffers.
    """
    import torch.distributed as dist
    from .api_low_level import NuFHECloudKey
    rank = dist.get_rank(group)
    meta = [cloud_key.params if rank == src else None]
    global_src = dist.get_global_rank(group, src) if group is not None else src
    dist.broadcast_object_list(meta, src=global_src, group=group)
    params = meta[0]
    if rank == src:
        image = cloud_key._native.export_image()
    else:
        from .bootstrap import NativeCloudKey
        probe = NativeCloudKey(thr, params.in_out_params.size, params._transform_type, params._tlwe_mask_size)
        image = torch.empty(probe.image_bytes(), dtype=torch.uint8, device=thr.device)
        probe.destroy()
    if dist.get_backend(group) == "gloo" and image.is_cuda:       # test route: ranks sharing a GPU, staged through the host
        staged = image.cpu()
        dist.broadcast(staged, src=global_src, group=group)
        if rank != src:
            image.copy_(staged)
    else:
        dist.broadcast(image, src=global_src, group=group)
    if rank == src:
        return cloud_key
    return NuFHECloudKey.from_device_image(thr, params, image)


def gather_threads(dst_thr, parts):
    """
    Single-process counterpart of the RCCL gather: the reference drives several GPUs from the threads of ONE process and
    its main thread collects the result slices (examples/multi_gpu.py:46-114).  ``parts`` = [(DeviceThread, LweSampleArray
    slice), ...] in slice order, each slice living on its thread's device; returns the concatenated LweSampleArray on
    ``dst_thr``'s device.  The copies are peer-to-peer (``nufhe_gather``: enqueued on each SOURCE stream behind the gate
    that produced the slice; the destination stream waits for them) -- nothing goes through the host.
    """
    import ctypes
    from . import _lib
    from .lwe import LweSampleArray
    if not parts:
        raise ValueError("nothing to gather")
    params = parts[0][1].params
    lead = [int(numpy_prod(ct.shape)) for _, ct in parts]
    total = sum(lead)
    n = params.size
    # allocated under the DESTINATION's stream: a recycled block's previous user is then work on that stream, which the
    # copies of nufhe_gather are ordered behind (its ordering contract, include/nufhe_hip.h)
    with torch.cuda.device(dst_thr.device), torch.cuda.stream(dst_thr._torch_stream):
        a = torch.empty((total, n), dtype=torch.int32, device=dst_thr.device)
        b = torch.empty((total,), dtype=torch.int32, device=dst_thr.device)
        cv = torch.empty((total,), dtype=torch.float32, device=dst_thr.device)
    keep = []
    for dst, field, row in ((a, 'a', 4 * n), (b, 'b', 4), (cv, 'current_variances', 4)):
        count = len(parts)
        srcs = (ctypes.c_void_p * count)()
        ptrs = (ctypes.c_void_p * count)()
        sizes = (ctypes.c_size_t * count)()
        offs = (ctypes.c_size_t * count)()
        pos = 0
        for i, (thr, ct) in enumerate(parts):
            if ct.params != params:
                raise ValueError("slices with different LWE parameters")
            src = getattr(ct, field)
            with torch.cuda.stream(thr._torch_stream):     # a packing copy belongs on the slice's own queue
                t = src.reshape((lead[i], n) if field == 'a' else (lead[i],)).contiguous()
            if t.data_ptr() != src.data_ptr():
                keep.append((thr, t))      # a strided slice was packed into a temporary: it must outlive its copy
            srcs[i] = thr.handle.value
            ptrs[i] = t.data_ptr() if lead[i] else None
            sizes[i] = lead[i] * row
            offs[i] = pos * row
            pos += lead[i]
        _lib.check(_lib.lib().nufhe_gather(dst_thr.handle, ctypes.c_void_p(dst.data_ptr()), offs, srcs, ptrs, sizes, count))
    for thr in {id(t): t for t, _ in keep}.values():
        with torch.cuda.stream(thr._torch_stream):
            thr.synchronize()          # (only when a temporary was made; contiguous slices are copied in place, no sync)
    return LweSampleArray(params, a, b, cv)


def numpy_prod(shape):
    out = 1
    for d in shape:
        out *= int(d)
    return out


def shard_bounds(nbits: int, world_size: int, rank: int):
    """Contiguous slice [lo, hi) of a flattened batch owned by ``rank`` (sizes differ by at most 1;
    the first ``nbits % world_size`` ranks get the extra bit)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d out of range [0, %d)" % (rank, world_size))
    base, extra = divmod(nbits, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_ciphertext(ciphertext, world_size: int, rank: int):
    """View of the rank's slice along the first axis of an LweSampleArray-like object."""
    lo, hi = shard_bounds(ciphertext.shape[0], world_size, rank)
    return ciphertext[lo:hi]


def _group_info(group, dst):
    """(world, group-local rank, group-local dst, global dst).  ``dst`` is GROUP-LOCAL everywhere in
    this module (rank order inside the group = slice order); torch's collectives want the global
    rank, so it is translated here once."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dst is None:
        return world, rank, None, None
    if not (0 <= dst < world):
        raise ValueError("dst %d out of range [0, %d)" % (dst, world))
    global_dst = dist.get_global_rank(group, dst) if group is not None else dst
    return world, rank, dst, global_dst


def _unpad(full, nbits, world, cap):
    """[world * cap, ...] gathered with ragged slices padded to ``cap`` -> [nbits, ...] in rank order"""
    if nbits == world * cap:
        return full
    pieces = []
    for r in range(world):
        l, h = shard_bounds(nbits, world, r)
        pieces.append(full[r * cap:r * cap + (h - l)])
    return torch.cat(pieces, dim=0)


def gather_arrays(local_tensors, nbits: int, group=None, dst=0):
    """
    Collects per-rank result slices (first axis = bits, possibly ragged by one) into full arrays, in
    rank order.  ``local_tensors`` is a tuple such as (a [b_r, n], b [b_r], cv [b_r]).

    ``dst`` = group-local rank that receives the result (default 0, the reference's main thread
    collecting the slices, examples/multi_gpu.py:104-107): returns a tuple of full tensors
    [nbits, ...] there and None on the other ranks -- every rank sends its 2008 bytes per bit exactly
    once.  ``dst=None`` gathers on every rank (world_size times the traffic; for callers that
    continue a circuit on all ranks).
    """
    import torch.distributed as dist
    world, rank, dst, global_dst = _group_info(group, dst)
    base, extra = divmod(nbits, world)
    cap = base + (1 if extra else 0)
    lo, hi = shard_bounds(nbits, world, rank)
    gloo_on_gpu = dist.get_backend(group) == 'gloo'
    out = []
    for t in local_tensors:
        if t.shape[0] != hi - lo:
            raise ValueError("local slice has %d bits, expected %d" % (t.shape[0], hi - lo))
        padded = t
        if t.shape[0] != cap:      # ragged tail: pad to the common capacity for the collective
            padded = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            padded[:t.shape[0]] = t
        padded = padded.contiguous()
        # test-only route (several ranks sharing one GPU over gloo): stage through the host
        stage = gloo_on_gpu and padded.is_cuda
        send = padded.cpu() if stage else padded
        full = None
        if dst is None:
            full = torch.empty((world * cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=send.device)
            dist.all_gather_into_tensor(full, send, group=group)
        elif rank == dst:
            full = torch.empty((world * cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=send.device)
            dist.gather(send, list(full.split(cap, dim=0)), dst=global_dst, group=group)
        else:
            dist.gather(send, None, dst=global_dst, group=group)
        if full is not None:
            if stage:
                full = full.to(t.device)
            full = _unpad(full, nbits, world, cap)
        out.append(full)
    if dst is not None and rank != dst:
        return None
    return tuple(out)


def gather_ciphertext(local_ct, nbits: int, group=None, dst=0):
    """Collects an LweSampleArray's slices; returns (a, b, current_variances) full tensors on ``dst``
    (every rank for ``dst=None``), None elsewhere."""
    return gather_arrays((local_ct.a, local_ct.b, local_ct.current_variances), nbits, group=group, dst=dst)


# ----------------------------------------------------------------------------------------------
# One buffer per result: a | b | variances as views of a single int32 allocation
# ----------------------------------------------------------------------------------------------

class PackedCiphertext:
    """
    Storage of one rank's result slice laid out for a single collective: an int32 buffer of
    ``capacity * (n + 2)`` words = ``a [capacity, n] | b [capacity] | variances [capacity]`` (the
    variances are float32 seen through an int32 view; same bytes).  ``ciphertext`` is an
    LweSampleArray over the first ``nbits`` rows -- pass it as ``dest=`` to a gate.
    ``capacity`` >= nbits is the common slice size of a ragged batch (the padding rows travel as
    zeros).
    """

    def __init__(self, params, nbits: int, device, capacity=None, sample_array_class=None):
        n = params.size
        capacity = nbits if capacity is None else capacity
        if capacity < nbits:
            raise ValueError("capacity %d < nbits %d" % (capacity, nbits))
        self.n = n
        self.nbits = nbits
        self.capacity = capacity
        self.flat = torch.zeros(capacity * (n + 2), dtype=torch.int32, device=device)
        a, b, cv = unpack_views(self.flat, capacity, n)
        if sample_array_class is None:
            from .lwe import LweSampleArray as sample_array_class
        self.ciphertext = sample_array_class(params, a[:nbits], b[:nbits], cv[:nbits])


def unpack_views(flat, capacity: int, n: int):
    """(a [capacity, n], b [capacity], variances [capacity] float32) views of one packed block"""
    a = flat[:capacity * n].view(capacity, n)
    b = flat[capacity * n:capacity * (n + 1)]
    cv = flat[capacity * (n + 1):capacity * (n + 2)].view(torch.float32)
    return a, b, cv


class AsyncGather:
    """Handle of a gather started by ``gather_packed_async``.

    ``wait()`` makes the caller's current stream wait for the collective (device-side for RCCL, so
    the host does not block) and returns, on the destination rank, (a, b, variances) full tensors
    [nbits_total, ...] in rank order; None on the other ranks.  The receive buffer holds one
    a | b | variances block per rank, so for more than one rank the three arrays are assembled with one
    ``torch.cat`` each (a copy of the gathered bytes on the current stream); only a single-rank group gets
    views of the receive buffer.  Callers that consume the packed blocks themselves (``bench.py``) pass
    ``unpack=False`` and pay nothing."""

    def __init__(self, work, recv, keep, world, cap, n, nbits_total, device, is_dst):
        self._work, self._recv, self._keep = work, recv, keep
        self._world, self._cap, self._n, self._nbits = world, cap, n, nbits_total
        self._device, self._is_dst = device, is_dst
        self._result = None
        self._done = False

    def wait(self, unpack=True):
        """``unpack=False``: only order the stream behind the collective (the send buffer may be reused)"""
        if not self._done:
            if self._work is not None:
                self._work.wait()
            self._done = True
            self._keep = None
        if unpack and self._result is None and self._is_dst:
            recv = self._recv if self._recv.device == self._device else self._recv.to(self._device)
            blocks = recv.view(self._world, self._cap * (self._n + 2))
            parts = [unpack_views(blocks[r], self._cap, self._n) for r in range(self._world)]
            if self._nbits == self._world * self._cap and self._world == 1:
                self._result = parts[0]
            else:
                res = []
                for k in range(3):
                    pieces = []
                    for r in range(self._world):
                        l, h = shard_bounds(self._nbits, self._world, r)
                        pieces.append(parts[r][k][:h - l])
                    res.append(torch.cat(pieces, dim=0))
                self._result = tuple(res)
        return self._result


def gather_packed_async(packed: PackedCiphertext, nbits_total: int, group=None, dst=0, recv=None):
    """
    Starts the gather of every rank's PackedCiphertext to group-local rank ``dst`` as ONE collective
    (``async_op=True``) and returns an :class:`AsyncGather`.  All ranks must use the same
    ``capacity`` = ceil(nbits_total / world).  ``recv`` (destination rank only) lets the caller
    reuse a receive buffer of ``world * capacity * (n + 2)`` int32 words.

    RCCL orders the collective behind the work already queued on the current stream (the gate that
    fills ``packed``) and runs it on its own stream; whatever the caller queues next overlaps it.
    The caller must not overwrite ``packed`` before ``wait()``.
    """
    import torch.distributed as dist
    world, rank, dst, global_dst = _group_info(group, dst)
    if dst is None:
        raise ValueError("the packed gather has one destination; use gather_arrays(dst=None) to gather everywhere")
    cap = -(-nbits_total // world)
    if packed.capacity != cap:
        raise ValueError("packed capacity %d, expected ceil(%d / %d) = %d" % (packed.capacity, nbits_total, world, cap))
    lo, hi = shard_bounds(nbits_total, world, rank)
    if packed.nbits != hi - lo:
        raise ValueError("local slice has %d bits, expected %d" % (packed.nbits, hi - lo))
    send = packed.flat
    device = send.device
    stage = dist.get_backend(group) == 'gloo' and send.is_cuda      # test-only route: through the host
    if stage:
        send = send.cpu()
    words = cap * (packed.n + 2)
    is_dst = rank == dst
    if is_dst:
        if recv is None or stage:
            recv = torch.empty(world * words, dtype=torch.int32, device=send.device)
        elif recv.numel() != world * words or recv.dtype != torch.int32:
            raise ValueError("receive buffer must hold %d int32 words" % (world * words))
        work = dist.gather(send, list(recv.split(words)), dst=global_dst, group=group, async_op=True)
    else:
        recv = None
        work = dist.gather(send, None, dst=global_dst, group=group, async_op=True)
    return AsyncGather(work, recv, send, world, cap, packed.n, nbits_total, device, is_dst)
