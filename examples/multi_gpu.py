"""
Multi-GPU NAND (equivalent of the reference's examples/multi_gpu.py, BASELINE config 4 shape).

The reference starts one Python thread per GPU and ships pickled keys / ciphertext slices through
host memory.  Here: one process per GPU, bits sharded contiguously, the cloud key replicated
(every rank loads the same serialized key), results gathered with one RCCL all_gather per array.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29511 examples/multi_gpu.py --bits 32768
"""
import argparse
import os
import sys

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
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ctx = nufhe.Context(rng=nufhe.DeterministicRNG(123), thread=DeviceThread(local_rank))
    # rank 0 plays the client: keys, plaintext, encryption; everything is broadcast as bytes
    if rank == 0:
        secret_key, cloud_key = ctx.make_key_pair()
        rs = numpy.random.RandomState(1)
        bits1 = rs.randint(0, 2, size=args.bits).astype(bool)
        bits2 = rs.randint(0, 2, size=args.bits).astype(bool)
        payload = [cloud_key.dumps(), ctx.encrypt(secret_key, bits1).dumps(), ctx.encrypt(secret_key, bits2).dumps()]
    else:
        payload = [None, None, None]
    if world > 1:
        dist.broadcast_object_list(payload, src=0)
    cloud_key = ctx.load_cloud_key(payload[0]) if rank != 0 else cloud_key
    ct1 = ctx.load_ciphertext(payload[1]); ct2 = ctx.load_ciphertext(payload[2])

    vm = ctx.make_virtual_machine(cloud_key)
    part = vm.gate_nand(multi_gpu.shard_ciphertext(ct1, world, rank), multi_gpu.shard_ciphertext(ct2, world, rank))
    if world > 1:
        a, b, cv = multi_gpu.gather_ciphertext(part, args.bits)
        result = nufhe.LweSampleArray(part.params, a, b, cv)
    else:
        result = part
    if rank == 0:
        assert (ctx.decrypt(secret_key, result) == ~(bits1 & bits2)).all()
        print("multi-GPU gate_nand OK: %d bits over %d GPU(s)" % (args.bits, world))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
