"""
N > 1 path on CPU: world_size-2 gloo processes exercise the sharding / gather layer that
bench.py --gpus N and the multi-GPU example use (nufhe_amd/multi_gpu.py).  The per-rank "gate" is
a stand-in integer function of the inputs (the real gate needs a GPU); what is tested is that the
shards partition the batch and that the gathered result equals the unsharded computation, for even
and ragged batch sizes.
"""

import os
import socket
import sys

import numpy
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_gate(a, b):
    # any deterministic int32 elementwise function of the two "ciphertexts"
    return (a * 3 - b * 5 + 7).to(torch.int32)


def _worker(rank, world, port, nbits, results):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nufhe_amd import multi_gpu
    g = torch.Generator().manual_seed(1234)     # same data on every rank (replicated inputs)
    a = torch.randint(-2**31, 2**31 - 1, (nbits, 500), dtype=torch.int32, generator=g)
    b = torch.randint(-2**31, 2**31 - 1, (nbits,), dtype=torch.int32, generator=g)
    a2 = torch.randint(-2**31, 2**31 - 1, (nbits, 500), dtype=torch.int32, generator=g)
    b2 = torch.randint(-2**31, 2**31 - 1, (nbits,), dtype=torch.int32, generator=g)
    lo, hi = multi_gpu.shard_bounds(nbits, world, rank)
    local = (_fake_gate(a[lo:hi], a2[lo:hi]), _fake_gate(b[lo:hi], b2[lo:hi]),
             torch.full((hi - lo,), float(rank)))
    full = multi_gpu.gather_arrays(local, nbits, dst=None)       # every rank receives everything
    ok = bool((full[0] == _fake_gate(a, a2)).all() and (full[1] == _fake_gate(b, b2)).all())
    on0 = multi_gpu.gather_arrays(local, nbits)                  # default: rank 0 only
    if rank == 0:
        ok = ok and all(bool((x == y).all()) for x, y in zip(on0, full))
    else:
        ok = ok and on0 is None
    owners = full[2].to(torch.int64)
    expect_owner = torch.cat([torch.full((multi_gpu.shard_bounds(nbits, world, r)[1]
                                          - multi_gpu.shard_bounds(nbits, world, r)[0],), r) for r in range(world)])
    ok = ok and bool((owners == expect_owner).all())
    results[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


class _Params:
    size = 500


class _Ct:
    def __init__(self, params, a, b, cv):
        self.params, self.a, self.b, self.current_variances = params, a, b, cv


def _packed_worker(rank, world, port, nbits, results):
    """the one-collective, asynchronous gather of bench.py / examples/multi_gpu.py: two result buffers in
    flight, each filled by a stand-in gate, gathered to rank 0 while the next one is being filled"""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nufhe_amd import multi_gpu
    g = torch.Generator().manual_seed(99)
    a = torch.randint(-2**31, 2**31 - 1, (nbits, 500), dtype=torch.int32, generator=g)
    b = torch.randint(-2**31, 2**31 - 1, (nbits,), dtype=torch.int32, generator=g)
    cv = torch.rand((nbits,), generator=g)
    lo, hi = multi_gpu.shard_bounds(nbits, world, rank)
    cap = -(-nbits // world)
    bufs = [multi_gpu.PackedCiphertext(_Params, hi - lo, 'cpu', capacity=cap, sample_array_class=_Ct) for _ in range(2)]
    ok = True
    pending = [None, None]
    outs = []
    for step in range(4):
        k = step % 2
        if pending[k] is not None:
            outs.append((step - 2, pending[k].wait()))         # buffer k is free again
        ct = bufs[k].ciphertext
        ct.a.copy_(_fake_gate(a[lo:hi], a[lo:hi] + step))       # the "gate" writes into the packed views
        ct.b.copy_(_fake_gate(b[lo:hi], b[lo:hi] + step))
        ct.current_variances.copy_(cv[lo:hi] + step)
        pending[k] = multi_gpu.gather_packed_async(bufs[k], nbits, dst=0)
    for k in ((4 % 2), (5 % 2)):
        outs.append((4 - 2 + (0 if k == 0 else 1), pending[k].wait()))
    for step, res in outs:
        if rank == 0:
            ok = ok and bool((res[0] == _fake_gate(a, a + step)).all()) and bool((res[1] == _fake_gate(b, b + step)).all())
            ok = ok and bool((res[2] == cv + step).all()) and res[0].shape == (nbits, 500) and res[2].dtype == torch.float32
        else:
            ok = ok and res is None
    # wrong capacity / wrong slice size are refused before any collective starts
    try:
        multi_gpu.gather_packed_async(multi_gpu.PackedCiphertext(_Params, hi - lo, 'cpu', capacity=cap + 1,
                                                                 sample_array_class=_Ct), nbits)
        ok = False
    except ValueError:
        pass
    results[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('nbits', [64, 37])
def test_packed_async_gather_world2(nbits):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_packed_worker, args=(world, port, nbits, results), nprocs=world, join=True)
    assert all(results.get(r) for r in range(world)), dict(results)


def _subgroup_worker(rank, world, port, results):
    """dst is GROUP-LOCAL: in a group made of global ranks (2, 1) the destination 0 is global rank 2"""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nufhe_amd import multi_gpu
    grp = dist.new_group(ranks=[1, 2])
    ok = True
    if rank in (1, 2):
        gr = dist.get_rank(grp)
        lo, hi = multi_gpu.shard_bounds(10, 2, gr)
        local = (torch.arange(lo, hi, dtype=torch.int32),)
        res = multi_gpu.gather_arrays(local, 10, group=grp, dst=1)      # group-local 1 = global rank 2
        if rank == 2:
            ok = res is not None and bool((res[0] == torch.arange(10, dtype=torch.int32)).all())
        else:
            ok = res is None
    results[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_gather_destination_is_group_local():
    world = 3
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_subgroup_worker, args=(world, port, results), nprocs=world, join=True)
    assert all(results.get(r) for r in range(world)), dict(results)


@pytest.mark.parametrize('nbits', [64, 37])
def test_shard_and_gather_world2(nbits):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, nbits, results), nprocs=world, join=True)
    assert all(results.get(r) for r in range(world)), dict(results)


def test_shard_bounds_partition():
    sys.path.insert(0, ROOT)
    from nufhe_amd.multi_gpu import shard_bounds
    for nbits in (0, 1, 7, 32, 4096, 32768, 1001):
        for world in (1, 2, 3, 8):
            bounds = [shard_bounds(nbits, world, r) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == nbits
            assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in bounds]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


_RANK_SCRIPT = '''
import os, sys
import torch, torch.distributed as dist
dist.init_process_group("gloo")
t = torch.tensor([dist.get_rank() + 1])
dist.all_reduce(t)
out = sys.argv[sys.argv.index("--out") + 1]
open(os.path.join(out, "rank%d" % dist.get_rank()), "w").write(
    "%d %s %d %s" % (dist.get_world_size(), os.environ["MASTER_ADDR"], int(t.item()), os.environ.get("OMP_NUM_THREADS")))
dist.destroy_process_group()
'''


@pytest.mark.parametrize('nproc', [2, 3])
def test_launch_ranks_starts_n_ranks_from_a_plain_process(tmp_path, nproc):
    """multi_gpu.launch_ranks -- what `python bench.py --gpus N` and `python examples/multi_gpu.py --gpus N` call when
    no launcher started them: N ranks on this node under torch.distributed.run, rendezvous on 127.0.0.1 and a free port,
    the host's threads shared out, the script's own arguments passed through, exit status returned."""
    from nufhe_amd import multi_gpu
    script = tmp_path / 'ranks.py'
    script.write_text(_RANK_SCRIPT)
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'OMP_NUM_THREADS'):
        os.environ.pop(k, None)
    try:
        rc = multi_gpu.launch_ranks(str(script), ['--out', str(tmp_path)], nproc, backend='gloo')
    finally:
        os.environ.clear(); os.environ.update(env)
    assert rc == 0
    for r in range(nproc):
        world, addr, total, omp = (tmp_path / ('rank%d' % r)).read_text().split()
        assert int(world) == nproc and addr == '127.0.0.1' and int(total) == nproc * (nproc + 1) // 2
        assert int(omp) == max(1, (os.cpu_count() or 1) // nproc)
    bad = tmp_path / 'bad.py'
    bad.write_text('import sys; sys.exit(3)')
    assert multi_gpu.launch_ranks(str(bad), [], 2, backend='gloo') != 0


def test_launch_ranks_refusals(monkeypatch):
    from nufhe_amd import multi_gpu
    monkeypatch.delenv('RANK', raising=False)
    if torch.cuda.device_count() < 2:
        with pytest.raises(RuntimeError, match='need 2 GPUs'):           # RCCL: one GPU per rank, checked before starting
            multi_gpu.launch_ranks('x.py', [], 2, backend='nccl')
    with pytest.raises(ValueError):
        multi_gpu.launch_ranks('x.py', [], 0, backend='gloo')
    monkeypatch.setenv('RANK', '0')
    with pytest.raises(RuntimeError, match='inside a rank'):
        multi_gpu.launch_ranks('x.py', [], 2, backend='gloo')


def test_bench_gpus_n_without_gpus_fails_loudly():
    """`python bench.py --gpus 2` where RCCL cannot give every rank a GPU (this container has none) exits non-zero
    with the reason and prints no JSON line -- it never degrades to a one-GPU measurement."""
    import subprocess
    if torch.cuda.device_count() >= 2:
        pytest.skip('two GPUs present')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'NUFHE_BENCH_BACKEND')}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                          capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert proc.returncode != 0 and 'need 2 GPUs' in proc.stderr and '{' not in proc.stdout
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                          capture_output=True, text=True, timeout=300, cwd=ROOT,
                          env=dict(env, RANK='0', WORLD_SIZE='4', LOCAL_RANK='0'))
    assert proc.returncode != 0 and 'WORLD_SIZE=4' in proc.stderr and '{' not in proc.stdout
