"""
Multi-GPU NAND (equivalent of the reference's examples/multi_gpu.py, BASELINE config 4 shape).

The reference starts one Python thread per GPU and ships pickled keys / ciphertext slices through
host memory (examples/multi_gpu.py:46-114).  Here: one process per GPU, bits sharded contiguously,
the cloud key replicated (every rank loads the same serialized key), and the result slices GATHERED
TO RANK 0 -- the reference's main thread collecting the parts (:104-107) -- as one RCCL collective
per gate (a | b | variances of a slice share one buffer, nufhe_amd.multi_gpu.PackedCiphertext).
Only rank 0 holds the gathered ciphertext and decrypts it; the other ranks get None back.

    python examples/multi_gpu.py --gpus 8 --bits 32768          # starts its own 8 ranks (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29511 examples/multi_gpu.py --gpus 8 --bits 32768       # the same, launched from outside

The cloud key reaches the other ranks as ONE device broadcast of its 98.6 MB image
(multi_gpu.broadcast_cloud_key: RCCL over xGMI); ``--via-host`` takes the reference's route instead
(the serialized key pickled through the host, every rank re-uploading it).

``--backend gloo`` is the test route (several ranks may share one GPU; slices are staged through
the host); the default ``nccl`` is RCCL over xGMI.
"""
import argparse
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC between the ranks of a node (RCCL)

import numpy
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nufhe_amd as nufhe
from nufhe_amd import multi_gpu
from nufhe_amd.device import DeviceThread


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", type=int, default=64)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=os.environ.get("NUFHE_BENCH_BACKEND", "nccl"))
    ap.add_argument("--gpus", type=int, default=None, help="number of ranks (= GPUs) to run on; started here if "
                    "this process was not launched by torch.distributed.run")
    ap.add_argument("--via-host", action="store_true", help="ship the cloud key as the reference does (pickled "
                    "through the host) instead of one device broadcast")
    args = ap.parse_args()
    if "RANK" not in os.environ and args.gpus is not None and args.gpus > 1:
        os.environ["NUFHE_BENCH_BACKEND"] = args.backend
        try:
            sys.exit(multi_gpu.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, backend=args.backend))
        except RuntimeError as e:
            sys.exit("multi_gpu.py: " + str(e))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is not None and "RANK" in os.environ and world != args.gpus:
        sys.exit("multi_gpu.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ          # launched by torch.distributed.run, with any world size
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    thr = DeviceThread(local_rank)
    ctx = nufhe.Context(rng=nufhe.DeterministicRNG(123), thread=thr)
    # rank 0 plays the client: keys, plaintext, encryption; everything is broadcast as bytes
    secret_key = cloud_key = None
    if rank == 0:
        secret_key, cloud_key = ctx.make_key_pair()
        rs = numpy.random.RandomState(1)
        bits1 = rs.randint(0, 2, size=args.bits).astype(bool)
        bits2 = rs.randint(0, 2, size=args.bits).astype(bool)
        payload = [cloud_key.dumps() if args.via_host else None,
                   ctx.encrypt(secret_key, bits1).dumps(), ctx.encrypt(secret_key, bits2).dumps()]
    else:
        payload = [None, None, None]
    if distributed and world > 1:
        dist.broadcast_object_list(payload, src=0)           # the ciphertexts (and, with --via-host, the key) as bytes
    if args.via_host:
        if rank != 0:
            cloud_key = ctx.load_cloud_key(payload[0])
    elif distributed:
        cloud_key = multi_gpu.broadcast_cloud_key(thr, cloud_key, src=0)     # one device collective
    ct1 = ctx.load_ciphertext(payload[1]); ct2 = ctx.load_ciphertext(payload[2])

    vm = ctx.make_virtual_machine(cloud_key)
    lo, hi = multi_gpu.shard_bounds(args.bits, world, rank)
    part1 = multi_gpu.shard_ciphertext(ct1, world, rank)
    part2 = multi_gpu.shard_ciphertext(ct2, world, rank)
    if distributed:
        # the gate writes straight into the buffer that is gathered
        packed = multi_gpu.PackedCiphertext(part1.params, hi - lo, thr.device, capacity=-(-args.bits // world))
        vm.gate_nand(part1, part2, dest=packed.ciphertext)
        gathered = multi_gpu.gather_packed_async(packed, args.bits, dst=0).wait()
        result = nufhe.LweSampleArray(part1.params, *gathered) if rank == 0 else None    # None on ranks != 0
    else:
        result = vm.gate_nand(part1, part2)
    if rank == 0:
        assert result.shape == (args.bits,)
        assert (ctx.decrypt(secret_key, result) == ~(bits1 & bits2)).all()
        print("multi-GPU gate_nand OK: %d bits over %d GPU(s)%s%s" % (
            args.bits, world, ", gathered to rank 0 over %s" % args.backend if distributed else "",
            (", key via host pickle" if args.via_host else ", key as one device broadcast") if distributed else ""))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
