"""
Heterogeneous gate batch (nufhe_gate_batch / VirtualMachine.gate_batch; SURVEY 8f row 4, circuit-level fusion -- the
reference's circuit is a gate-by-gate chain, nufhe/operators_integer.py:64-95): a list of independent gates of different
kinds and sizes as ONE bootstrap launch.  Every job's result words (a, b, variances) are compared with the CPU oracle on
the same inputs and with the individual gate calls; plus the round-4 advisor items that live next to it in api.hip:
scratch pinning for captured graphs, the key-image header, the ordering of nufhe_gather.
"""
import ctypes
import os
import time

import numpy
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env(orc, oracle_keys):
    import gpu_helpers as H
    from nufhe_amd.device import DeviceThread
    import nufhe_amd
    thr = DeviceThread(0)
    lwe_key, tlwe_key, ck = oracle_keys
    cloud_key = H.cloud_key_from_arrays(thr, ck)
    secret_key = H.secret_key_from_array(thr, lwe_key)
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(456), thread=thr)
    vm = ctx.make_virtual_machine(cloud_key)
    return dict(H=H, thr=thr, ctx=ctx, vm=vm, ck=ck, lwe_key=lwe_key, cloud_key=cloud_key, secret_key=secret_key)


TRUTH = {
    'gate_nand': lambda a, b: ~(a & b), 'gate_or': lambda a, b: a | b, 'gate_and': lambda a, b: a & b,
    'gate_xor': lambda a, b: a ^ b, 'gate_xnor': lambda a, b: ~(a ^ b), 'gate_nor': lambda a, b: ~(a | b),
    'gate_andny': lambda a, b: ~a & b, 'gate_andyn': lambda a, b: a & ~b, 'gate_orny': lambda a, b: ~a | b,
    'gate_oryn': lambda a, b: a | ~b,
}


def _make(env, orc, rng, n):
    m = rng.uniform_bool((n,)).astype(bool)
    c = orc.encrypt(rng, env['lwe_key'], m)
    return m, c, env['H'].ciphertext_from_arrays(env['thr'], c)


def _oracle(orc, ck, name, cs):
    if name == 'gate_mux':
        return orc.gate_mux(ck, cs[0], cs[1], cs[2])
    return orc.gate(name, ck, cs[0], cs[1])


def _check_job(env, orc, name, result, inputs, upto=None):
    H = env['H']
    ra, rb, rcv = H.ct_arrays(result)
    ms = [m for m, _, _ in inputs]
    n = ra.shape[0] if upto is None else min(upto, ra.shape[0])
    exp = _oracle(orc, env['ck'], name, [tuple(x[:n] for x in c) for _, c, _ in inputs])
    assert (ra[:n] == exp[0]).all() and (rb[:n] == exp[1]).all() and (rcv[:n] == exp[2]).all(), name
    truth = numpy.where(ms[0], ms[1], ms[2]) if name == 'gate_mux' else TRUTH[name](ms[0], ms[1])
    assert (env['ctx'].decrypt(env['secret_key'], result) == truth).all(), name


def test_batch_of_mixed_gates_every_word_vs_oracle(env, orc):
    """binary | MUX | binary with odd sizes (one bit, a prime, zero), a MUX between two binary jobs"""
    rng = orc.DeterministicRNG(2025)
    spec = [('gate_nand', 5), ('gate_mux', 7), ('gate_xor', 1), ('gate_and', 0), ('gate_mux', 3), ('gate_oryn', 13)]
    jobs, keep = [], []
    for name, n in spec:
        ins = [_make(env, orc, rng, n) for _ in range(3 if name == 'gate_mux' else 2)]
        keep.append((name, ins))
        jobs.append((name,) + tuple(d for _, _, d in ins))
    results = env['vm'].gate_batch(jobs)
    assert len(results) == len(spec)
    for (name, ins), res in zip(keep, results):
        assert res.shape == (ins[0][0].shape[0],)
        if res.shape[0]:
            _check_job(env, orc, name, res, ins)
        # the individual call writes the same words
        single = getattr(env['vm'], name)(*[d for _, _, d in ins])
        assert single == res


def test_batch_all_ten_binary_gates_and_destinations(env, orc):
    """one launch holding every binary gate kind; results into caller-provided destinations, one of them a strided view"""
    rng = orc.DeterministicRNG(7)
    vm = env['vm']
    a = _make(env, orc, rng, 6); b = _make(env, orc, rng, 6)
    names = sorted(TRUTH)
    big = vm.empty_ciphertext((len(names), 12))
    jobs = [(name, a[2], b[2], big[i, ::2]) for i, name in enumerate(names)]
    results = vm.gate_batch(jobs)
    for i, name in enumerate(names):
        _check_job(env, orc, name, big[i, ::2].copy(), [a, b])
        assert results[i] == big[i, ::2]


@pytest.mark.parametrize('sizes', [(64, 64, 64, 64), (300, 300), (1500, 700, 200)])
def test_batch_crosses_the_kernel_families(env, orc, sizes):
    """total row counts that select the 8-wave team kernel (<= CUs), the pair kernel and the wave kernel + matrix-core
    keyswitch: first 16 bits of every job against the oracle, everything against the individual gate calls"""
    rng = orc.DeterministicRNG(sum(sizes))
    vm = env['vm']
    names = ['gate_nand', 'gate_mux', 'gate_xnor', 'gate_andny']
    jobs, keep = [], []
    for name, n in zip(names, sizes):
        ins = [_make(env, orc, rng, n) for _ in range(3 if name == 'gate_mux' else 2)]
        keep.append((name, ins))
        jobs.append((name,) + tuple(d for _, _, d in ins))
    results = vm.gate_batch(jobs)
    for (name, ins), res in zip(keep, results):
        _check_job(env, orc, name, res, ins, upto=16)
        assert getattr(vm, name)(*[d for _, _, d in ins]) == res


def test_batch_broadcast_operand_and_stepwise_mode(env, orc):
    """a one-bit operand broadcast over a job; and the reference's multi-kernel mode runs the same list gate by gate"""
    import nufhe_amd
    rng = orc.DeterministicRNG(11)
    vm = env['vm']
    one = _make(env, orc, rng, 1); many = _make(env, orc, rng, 9); other = _make(env, orc, rng, 4)
    r0, r1 = vm.gate_batch([('gate_and', one[2], many[2]), ('gate_nor', other[2], other[2])])
    assert r0.shape == (9,) and r1.shape == (4,)
    assert r0 == vm.gate_and(one[2], many[2]) and r1 == vm.gate_nor(other[2], other[2])
    pp = nufhe_amd.PerformanceParameters(env['cloud_key'].params, single_kernel_bootstrap=False)
    vm2 = env['ctx'].make_virtual_machine(env['cloud_key'], perf_params=pp)
    s0, s1 = vm2.gate_batch([('gate_and', one[2], many[2]), ('gate_nor', other[2], other[2])])
    assert s0 == r0 and s1 == r1


def test_batch_refusals(env, orc):
    rng = orc.DeterministicRNG(3)
    vm = env['vm']
    a = _make(env, orc, rng, 4); b = _make(env, orc, rng, 5)
    with pytest.raises(ValueError):
        vm.gate_batch([('gate_nand', a[2], b[2])])             # shapes do not broadcast
    with pytest.raises(ValueError):
        vm.gate_batch([('gate_not', a[2], a[2])])              # not a bootstrapped gate
    with pytest.raises(ValueError):
        vm.gate_batch([('gate_mux', a[2], a[2])])              # arity
    assert vm.gate_batch([]) == []
    # raw C boundary: unknown kind, wrong operand size
    from nufhe_amd import _lib
    from nufhe_amd.lwe import _Flat
    res = vm.empty_ciphertext((4,))
    job = (_lib.NufheGateJob * 1)()
    job[0].kind = 7; job[0].nbits = 4
    job[0].result = _Flat(res, (4,), output=True).desc; job[0].a = _Flat(a[2], (4,)).desc; job[0].b = _Flat(a[2], (4,)).desc
    with pytest.raises(ValueError, match='unknown kind'):
        _lib.call("nufhe_gate_batch", env['thr'].handle, env['cloud_key']._native.handle, job, 1, 1 << 29)
    job[0].kind = 0
    job[0].a.size = 499
    with pytest.raises(ValueError, match='LWE size'):
        _lib.call("nufhe_gate_batch", env['thr'].handle, env['cloud_key']._native.handle, job, 1, 1 << 29)


def test_batch_refuses_mux_with_another_mu_and_overlapping_results(env, orc):
    """round-5 advisor: the MUX fold is written for mu = 2^29; two jobs writing the same rows race in the finalize"""
    from nufhe_amd import _lib
    from nufhe_amd.lwe import _Flat
    rng = orc.DeterministicRNG(5)
    vm = env['vm']
    a = _make(env, orc, rng, 4)
    res = vm.empty_ciphertext((8,))
    jobs = (_lib.NufheGateJob * 2)()
    for j in range(2):
        jobs[j].kind = _lib.JOB_MUX if j == 0 else _lib.JOB_BINARY
        jobs[j].nbits = 4
        jobs[j].result = _Flat(res[4 * j:4 * j + 4], (4,), output=True).desc
        jobs[j].a = jobs[j].b = jobs[j].c = _Flat(a[2], (4,)).desc
        jobs[j].pa = jobs[j].pb = 1
    h = (env['thr'].handle, env['cloud_key']._native.handle)
    with pytest.raises(ValueError, match='mu = 2\\^29'):
        _lib.call("nufhe_gate_batch", *h, jobs, 2, 1 << 28)
    _lib.call("nufhe_gate_batch", *h, jobs, 2, 1 << 29)               # disjoint halves of one array: fine
    jobs[1].result = _Flat(res[2:6], (4,), output=True).desc
    with pytest.raises(ValueError, match='result views overlap'):
        _lib.call("nufhe_gate_batch", *h, jobs, 2, 1 << 29)
    # rows of one array interleaved (equal strides, a row apart): disjoint, accepted
    two = vm.empty_ciphertext((4, 2))
    jobs[0].result = _Flat(two[:, 0], (4,), output=True).desc
    jobs[1].result = _Flat(two[:, 1], (4,), output=True).desc
    if jobs[0].result.a_stride == jobs[1].result.a_stride == 2 * 500:    # (a strided view passed without a temporary)
        _lib.call("nufhe_gate_batch", *h, jobs, 2, 1 << 29)


def test_batch_operand_aliasing_is_the_same_in_both_modes(env, orc):
    """round-5 advisor: a job that reads another job's result sees the value from BEFORE the call, fused or gate by gate"""
    import nufhe_amd
    rng = orc.DeterministicRNG(17)
    vm = env['vm']
    pp = nufhe_amd.PerformanceParameters(env['cloud_key'].params, single_kernel_bootstrap=False)
    vm2 = env['ctx'].make_virtual_machine(env['cloud_key'], perf_params=pp)
    x = _make(env, orc, rng, 6); y = _make(env, orc, rng, 6)
    outs = []
    for machine in (vm, vm2):
        r = x[2].copy()                    # job 0 overwrites r, job 1 reads r
        s = machine.empty_ciphertext((6,))
        from nufhe_amd.gates import gate_batch
        gate_batch(env['thr'], env['cloud_key'], [('gate_nand', r, r, y[2]), ('gate_xor', s, r, y[2])],
                   perf_params=machine.perf_params)
        outs.append((r, s))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    assert (env['ctx'].decrypt(env['secret_key'], outs[0][1]) == (x[0] ^ y[0])).all()       # the OLD r = x


def test_host_allocation_failure_is_an_error_code_not_an_abort(tmp_path):
    """std::bad_alloc inside nufhe_ks_upload (a 49 MB host staging vector) must come back as
    NUFHE_ENOMEM / MemoryError through the extern "C" boundary, not as std::terminate."""
    import subprocess
    import sys
    script = tmp_path / "enomem.py"
    script.write_text('''
import ctypes, resource, sys, numpy
sys.path.insert(0, %r)
from nufhe_amd import _lib
L = _lib.lib()
ctx = ctypes.c_void_p(); key = ctypes.c_void_p()
_lib.check(L.nufhe_ctx_create(0, None, 1, ctypes.byref(ctx)))
_lib.check(L.nufhe_cloudkey_create(ctx, 500, 0, 1, ctypes.byref(key)))
rows = 1024 * 8
a = numpy.zeros((rows, 4, 500), numpy.int32); b = numpy.zeros((rows, 4), numpy.int32); cv = numpy.zeros((rows, 4), numpy.float32)
vm = 0
for line in open("/proc/self/status"):
    if line.startswith("VmSize:"):
        vm = int(line.split()[1]) * 1024
resource.setrlimit(resource.RLIMIT_AS, (vm + (8 << 20), resource.RLIM_INFINITY))
rc = L.nufhe_ks_upload(key, a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), cv.ctypes.data_as(ctypes.c_void_p))
resource.setrlimit(resource.RLIMIT_AS, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
print("rc", rc, L.nufhe_last_error().decode())
try:
    _lib.check(rc)
except MemoryError as e:
    print("MemoryError", e)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rc -5" in out.stdout and "MemoryError" in out.stdout and "bad_alloc" in out.stdout, out.stdout + out.stderr


def test_batch_of_small_gates_takes_the_time_of_one(env, orc):
    """the point of the entry point: four independent 64-bit gates of different kinds in about the time of one (each is
    a 500-step dependent chain at one bit per CU); measured here, asserted loosely (< 1.6 x one gate, vs ~4 x)"""
    rng = orc.DeterministicRNG(4)
    vm, thr = env['vm'], env['thr']
    names = ['gate_nand', 'gate_or', 'gate_xor', 'gate_andyn']
    ins = [[_make(env, orc, rng, 64) for _ in range(2)] for _ in names]
    dests = [vm.empty_ciphertext((64,)) for _ in names]
    jobs = [(n, i[0][2], i[1][2], d) for n, i, d in zip(names, ins, dests)]

    def timed(fn, reps=5):
        fn(); thr.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        thr.synchronize()
        return (time.perf_counter() - t0) / reps
    t_one = timed(lambda: vm.gate_nand(ins[0][0][2], ins[0][1][2], dest=dests[0]))
    t_seq = timed(lambda: [getattr(vm, n)(i[0][2], i[1][2], dest=d) for n, i, d in zip(names, ins, dests)])
    t_batch = timed(lambda: vm.gate_batch(jobs))
    print("one 64-bit gate %.2f ms, four in sequence %.2f ms, four as one batch %.2f ms" % (
        1e3 * t_one, 1e3 * t_seq, 1e3 * t_batch))
    assert t_batch < 1.6 * t_one and t_batch < 0.5 * t_seq
    for n, i, d in zip(names, ins, dests):
        _check_job(env, orc, n, d, i, upto=8)


def test_graph_survives_larger_eager_gates(orc):
    """round-4 advisor: a captured graph points into the context's grow-only scratch; a LARGER eager gate afterwards used
    to free it.  With the scratch pinned by the GateGraph the old buffers stay alive: capture at 3 bits, run 600 and 2100
    bits eagerly (both reallocate), replay the 3-bit graph -> still the eager result."""
    import torch
    import nufhe_amd as nufhe
    from nufhe_amd.device import DeviceThread
    stream = torch.cuda.Stream()
    rs = numpy.random.RandomState(5)
    with torch.cuda.stream(stream):
        thr = DeviceThread(0)
        ctx = nufhe.Context(rng=nufhe.DeterministicRNG(42), thread=thr)
        secret, cloud = ctx.make_key_pair()
        vm = ctx.make_virtual_machine(cloud)
        ms = [rs.randint(0, 2, 3).astype(bool) for _ in range(3)]
        cs = [ctx.encrypt(secret, m) for m in ms]

        def circuit():
            return vm.gate_mux(vm.gate_nand(cs[0], cs[1]), cs[2], cs[0])
        g = nufhe.GateGraph(thr)
        out = g.capture(circuit)
        g.replay()
        expect = circuit()
        assert out == expect
        for B in (600, 2100):
            big = [ctx.encrypt(secret, rs.randint(0, 2, B).astype(bool)) for _ in range(3)]
            vm.gate_mux(big[0], big[1], big[2])
        thr.synchronize()
        out.a.zero_(); out.b.zero_()
        g.replay()
        assert out == expect
        g.close()
        # growing DURING a capture is refused, not silently wrong
        huge = [ctx.encrypt(secret, rs.randint(0, 2, 5000).astype(bool)) for _ in range(2)]
        with pytest.raises(Exception):
            _capture_without_warmup(thr, lambda: vm.gate_nand(huge[0], huge[1]))
        thr.synchronize()
        assert vm.gate_nand(cs[0], cs[1]) == vm.gate_nand(cs[0], cs[1])      # the context still works
    # capture / replay / close called from OUTSIDE the stream's context (they enter it themselves; pinning is not tied to it)
    g3 = nufhe.GateGraph(thr)
    out3 = g3.capture(circuit)
    g3.replay()
    stream.synchronize()
    with torch.cuda.stream(stream):
        assert out3 == expect
    g3.close()
    g3.close()                                                               # idempotent


def _capture_without_warmup(thr, circuit):
    import torch
    from nufhe_amd import _lib
    stream = thr._torch_stream
    _lib.call("nufhe_ctx_pin_scratch", thr.handle, 1)
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(graph, stream=stream):
                circuit()
    finally:
        _lib.call("nufhe_ctx_pin_scratch", thr.handle, -1)


def test_key_image_header_refuses_a_foreign_image(env):
    """round-4 advisor: NTT and FFT key images have the same size; the header tells them apart at the C boundary"""
    import torch
    import nufhe_amd
    from nufhe_amd import _lib
    from nufhe_amd.api_low_level import NuFHEParameters
    from nufhe_amd.bootstrap import NativeCloudKey
    from nufhe_amd.device import ptr
    thr = env['thr']
    params, image = env['cloud_key'].device_image()
    assert bytes(image[:8].cpu().numpy().tobytes()) == b'NUFHEIMG'
    fft = NativeCloudKey(thr, params.in_out_params.size, 'FFT', 1)
    assert fft.image_bytes() == image.numel()
    with pytest.raises(ValueError, match='cloud-key image holds'):
        fft.import_image(image)
    k2 = NativeCloudKey(thr, params.in_out_params.size, 'NTT', 2)
    with pytest.raises(ValueError):
        k2.import_image(image)
    junk = torch.zeros_like(image)
    ntt = NativeCloudKey(thr, params.in_out_params.size, 'NTT', 1)
    with pytest.raises(ValueError, match='not a cloud-key image'):
        ntt.import_image(junk)
    stale = image.clone()
    stale[8:12] = torch.tensor([3, 0, 0, 0], dtype=torch.uint8, device=stale.device)      # another ABI version
    with pytest.raises(ValueError, match='ABI version 3'):
        ntt.import_image(stale)
    ntt.import_image(image)                                                                # the real one loads
    for k in (fft, k2, ntt):
        k.destroy()


def test_gather_waits_for_work_queued_on_the_destination(env, orc):
    """round-4 advisor: nufhe_gather's copies run on the SOURCE streams; they must wait for what is already queued on
    the destination's stream (the destination block may be recycled while its last user is still in flight).  A long
    kernel chain that WRITES the destination buffer is queued on dst's stream, then the gather: the gathered bytes must
    survive."""
    import torch
    import nufhe_amd
    from nufhe_amd import _lib
    from nufhe_amd.device import DeviceThread
    s_dst, s_src = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s_dst):
        dst_thr = DeviceThread(0)
    with torch.cuda.stream(s_src):
        src_thr = DeviceThread(0)
        src = torch.arange(1 << 20, dtype=torch.int32, device='cuda')
        src_thr.synchronize()
    with torch.cuda.stream(s_dst):
        dst = torch.empty(1 << 20, dtype=torch.int32, device='cuda')
        big = torch.ones(1 << 26, dtype=torch.float32, device='cuda')
        for _ in range(40):
            big.mul_(1.0000001)                       # tens of milliseconds of queued work on dst's stream ...
        dst.fill_(-1)                                 # ... whose last kernel writes the destination buffer
    offs = (ctypes.c_size_t * 1)(0)
    srcs = (ctypes.c_void_p * 1)(src_thr.handle.value)
    ptrs = (ctypes.c_void_p * 1)(src.data_ptr())
    sizes = (ctypes.c_size_t * 1)(4 << 20)
    _lib.check(_lib.lib().nufhe_gather(dst_thr.handle, ctypes.c_void_p(dst.data_ptr()), offs, srcs, ptrs, sizes, 1))
    with torch.cuda.stream(s_dst):
        got = dst.clone()
        dst_thr.synchronize()
    assert (got == src).all()
    assert torch.cuda.current_device() == 0


def test_batch_of_many_one_bit_jobs_and_table_ring(env, orc):
    """70 one-bit jobs of alternating kinds in one call (the job tables outgrow their first 4 KiB slot: the slot is
    reallocated), then six more calls back to back without a synchronisation in between (the ring of four table slots
    wraps: a slot's staging buffer is rewritten only after its previous upload has executed) -- every result equals the
    individual gate call, the first batch also the oracle."""
    rng = orc.DeterministicRNG(70)
    vm, thr = env['vm'], env['thr']
    a = _make(env, orc, rng, 70); b = _make(env, orc, rng, 70); c = _make(env, orc, rng, 70)
    kinds = ['gate_nand', 'gate_mux', 'gate_xor', 'gate_orny', 'gate_mux', 'gate_nor', 'gate_and']
    jobs = []
    for i in range(70):
        k = kinds[i % len(kinds)]
        ops = (a[2][i:i + 1], b[2][i:i + 1]) + ((c[2][i:i + 1],) if k == 'gate_mux' else ())
        jobs.append((k,) + ops)
    res = vm.gate_batch(jobs)
    for i, r in enumerate(res):
        k = kinds[i % len(kinds)]
        if i < 14:                                    # two of every kind against the oracle, the rest against the single gates
            ins = [(x[0][i:i + 1], tuple(v[i:i + 1] for v in x[1]), None) for x in ((a, b, c) if k == 'gate_mux' else (a, b))]
            _check_job(env, orc, k, r, ins)
        else:
            ops = (a[2][i:i + 1], b[2][i:i + 1]) + ((c[2][i:i + 1],) if k == 'gate_mux' else ())
            assert r == getattr(vm, k)(*ops), (i, k)
    outs = []
    for rep in range(6):                              # no synchronisation between the calls
        outs.append(vm.gate_batch([(kinds[(rep + j) % 7],) + ((a[2][:5], b[2][:5], c[2][:5]) if kinds[(rep + j) % 7] == 'gate_mux'
                                                             else (a[2][:5], b[2][:5])) for j in range(3 + rep)]))
    thr.synchronize()
    for rep, group in enumerate(outs):
        for j, r in enumerate(group):
            k = kinds[(rep + j) % 7]
            ops = (a[2][:5], b[2][:5], c[2][:5]) if k == 'gate_mux' else (a[2][:5], b[2][:5])
            assert r == getattr(vm, k)(*ops), (rep, j, k)


def test_batch_fuzz_against_individual_gate_calls(env, orc):
    """Random job lists -- 1 to 6 jobs, any of the eleven bootstrapped gates, sizes from empty to beyond the team limit,
    one-bit operands broadcast over a job, results written in place over an operand -- against the individual gate calls
    on the same inputs (which the other tests pin to the oracle): equal words every time."""
    import torch
    rs = numpy.random.RandomState(20250926)
    rng = orc.DeterministicRNG(99)
    vm = env['vm']
    names = sorted(TRUTH) + ['gate_mux']
    sizes = [0, 1, 2, 3, 17, 64, 100, 257]
    pool = {n: [_make(env, orc, rng, n)[2] for _ in range(3)] for n in sizes if n}
    one = [_make(env, orc, rng, 1)[2] for _ in range(3)]
    for trial in range(14):
        jobs, expect = [], []
        for _ in range(rs.randint(1, 7)):
            name = names[rs.randint(len(names))]
            n = sizes[rs.randint(len(sizes))]
            arity = 3 if name == 'gate_mux' else 2
            if n == 0:
                ops = [pool[1][k][:0] for k in range(arity)]
            else:
                ops = [pool[n][k] for k in range(arity)]
                if n > 1 and rs.rand() < 0.3:
                    ops[rs.randint(arity)] = one[rs.randint(3)]          # broadcast operand
            ops = [o.copy() for o in ops]                                # private buffers: some jobs overwrite an operand
            expect.append(getattr(vm, name)(*ops))
            if n > 0 and rs.rand() < 0.3:
                full = [k for k, o in enumerate(ops) if o.shape == expect[-1].shape]
                dest = ops[full[rs.randint(len(full))]]                  # result in place over one of the operands
                jobs.append((name,) + tuple(ops) + (dest,))
            else:
                jobs.append((name,) + tuple(ops))
        got = vm.gate_batch(jobs)
        torch.cuda.synchronize()
        for j, (g, e) in enumerate(zip(got, expect)):
            assert g == e, (trial, j, jobs[j][0], tuple(e.shape))


def test_gate_batch_inside_a_captured_graph(orc):
    """gate_batch is capturable: its job tables travel inside kernel arguments, so a recorded circuit made of batches
    (uint_min_many: 18 batches of three comparators) replays with new inputs written in place and gives the eager words."""
    import torch
    import nufhe_amd as nufhe
    from nufhe_amd.device import DeviceThread
    from nufhe_amd.operators_integer import uint_min_many, uintarray_to_bitarray, bitarray_to_uintarray
    stream = torch.cuda.Stream()
    rs = numpy.random.RandomState(8)
    with torch.cuda.stream(stream):
        thr = DeviceThread(0)
        ctx = nufhe.Context(rng=nufhe.DeterministicRNG(43), thread=thr)
        secret, cloud = ctx.make_key_pair()
        shapes = [(2,), (1,), (3,)]

        def fresh():
            xs = [rs.randint(0, 2**8, size=s).astype(numpy.uint8) for s in shapes]
            ys = [rs.randint(0, 2**8, size=s).astype(numpy.uint8) for s in shapes]
            return xs, ys
        xs, ys = fresh()
        ca = [ctx.encrypt(secret, uintarray_to_bitarray(x)) for x in xs]
        cb = [ctx.encrypt(secret, uintarray_to_bitarray(y)) for y in ys]
        outs = [nufhe.empty_ciphertext(thr, cloud.params, s + (8,)) for s in shapes]

        def circuit():
            uint_min_many(thr, cloud, outs, ca, cb)
            return outs
        g = nufhe.GateGraph(thr)
        g.capture(circuit)
        g.replay()
        thr.synchronize()
        for k in range(3):
            assert (bitarray_to_uintarray(ctx.decrypt(secret, outs[k])) == numpy.minimum(xs[k], ys[k])).all()
        first = [o.copy() for o in outs]
        xs, ys = fresh()                                   # new inputs IN PLACE, eager reference first
        for k in range(3):
            ca[k][...] = ctx.encrypt(secret, uintarray_to_bitarray(xs[k]))
            cb[k][...] = ctx.encrypt(secret, uintarray_to_bitarray(ys[k]))
        expect = [o.copy() for o in circuit()]
        assert not all(e == f for e, f in zip(expect, first))
        for o in outs:
            o.a.zero_()
        g.replay()
        thr.synchronize()
        for k in range(3):
            assert outs[k] == expect[k]
            assert (bitarray_to_uintarray(ctx.decrypt(secret, outs[k])) == numpy.minimum(xs[k], ys[k])).all()
        g.close()
