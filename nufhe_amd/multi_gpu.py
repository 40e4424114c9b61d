"""
Multi-GPU data parallelism over independent ciphertext bits (reference: examples/multi_gpu.py:46-114,
which ships pickled slices between Python threads through host memory).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU
tests).  The batch is cut into contiguous per-rank slices, the cloud key is replicated, every rank
runs its gates locally with no data-path collective, and the only communication is the result
gather to one rank (RCCL gather = grouped send/recv over xGMI).  The functions below contain no device-specific code: they work on whatever tensors
(``a``, ``b``, ``current_variances``) they are given.
"""

import torch


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


def gather_arrays(local_tensors, nbits: int, group=None, dst=0):
    """
    Collects per-rank result slices (first axis = bits, possibly ragged by one) into full arrays, in
    rank order.  ``local_tensors`` is a tuple such as (a [b_r, n], b [b_r], cv [b_r]).

    ``dst`` = rank that receives the result (default 0, the reference's main thread collecting the
    slices, examples/multi_gpu.py:104-107): returns a tuple of full tensors [nbits, ...] there and None
    on the other ranks -- every rank sends its 2008 bytes per bit exactly once.  ``dst=None`` gathers on
    every rank (world_size times the traffic; for callers that continue a circuit on all ranks).
    """
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
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
            dist.gather(send, list(full.split(cap, dim=0)), dst=dst, group=group)
        else:
            dist.gather(send, None, dst=dst, group=group)
        if full is not None:
            if stage:
                full = full.to(t.device)
            if extra:
                pieces = []
                for r in range(world):
                    l, h = shard_bounds(nbits, world, r)
                    pieces.append(full[r * cap:r * cap + (h - l)])
                full = torch.cat(pieces, dim=0)
        out.append(full)
    if dst is not None and rank != dst:
        return None
    return tuple(out)


def gather_ciphertext(local_ct, nbits: int, group=None, dst=0):
    """Collects an LweSampleArray's slices; returns (a, b, current_variances) full tensors on ``dst``
    (every rank for ``dst=None``), None elsewhere."""
    return gather_arrays((local_ct.a, local_ct.b, local_ct.current_variances), nbits, group=group, dst=dst)
