"""
Gate circuits captured into one hipGraph (nufhe_amd/graph.py; SURVEY 8f row 4 "graph capture", the reference's circuit
example nufhe/operators_integer.py:64-95): the replayed circuit writes the same ciphertext words as the eager calls, for
new inputs written into the captured input buffers in place.
"""
import numpy
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def side_stream_env():
    import torch
    import nufhe_amd as nufhe
    from nufhe_amd.device import DeviceThread
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        thr = DeviceThread(0)
        ctx = nufhe.Context(rng=nufhe.DeterministicRNG(42), thread=thr)
        secret, cloud = ctx.make_key_pair()
        vm = ctx.make_virtual_machine(cloud)
    return dict(stream=stream, thr=thr, ctx=ctx, secret=secret, cloud=cloud, vm=vm)


def test_default_stream_is_refused():
    import nufhe_amd as nufhe
    from nufhe_amd.device import DeviceThread
    with pytest.raises(ValueError, match='default stream'):
        nufhe.GateGraph(DeviceThread(0))


def test_captured_gate_chain_equals_eager(side_stream_env):
    """NAND -> MUX -> XOR with a constant in between (every kind of launch a circuit makes: fused binary gate, two-job MUX,
    device-side constant fill, matrix-core and LDS keyswitch sizes), sizes 3 (8-wave team kernel) and 600 (pair kernel)."""
    import torch
    import nufhe_amd as nufhe
    e = side_stream_env
    ctx, vm, secret, stream = e['ctx'], e['vm'], e['secret'], e['stream']
    rs = numpy.random.RandomState(1)
    for B in (3, 600):
        with torch.cuda.stream(stream):
            ms = [rs.randint(0, 2, B).astype(bool) for _ in range(3)]
            cs = [ctx.encrypt(secret, m) for m in ms]

            def circuit():
                t = vm.gate_nand(cs[0], cs[1])
                one = vm.empty_ciphertext((B,))
                vm.gate_constant(True, dest=one)
                u = vm.gate_mux(t, cs[2], one)
                return vm.gate_xor(u, cs[0])
            g = nufhe.GateGraph(e['thr'])
            out = g.capture(circuit)                           # records; the output buffers are filled by replay()
            g.replay()
            first = out.copy()
            assert first == circuit()                          # replay == eager, same inputs
            # new inputs, in place; the eager reference is computed BEFORE the replay overwrites `out`
            ms2 = [rs.randint(0, 2, B).astype(bool) for _ in range(3)]
            for c, m in zip(cs, ms2):
                c[...] = ctx.encrypt(secret, m)
            expect = circuit()
            assert not (expect == first)
            g.replay()
            assert out == expect
            t = ~(ms2[0] & ms2[1])
            assert (ctx.decrypt(secret, out) == (numpy.where(t, ms2[2], True) ^ ms2[0])).all()
            # replaying twice gives the same words again (nothing in the graph depends on its own previous output)
            g.replay()
            assert out == expect


def test_captured_uint_min(side_stream_env):
    """The reference's circuit example (operators_integer.py:64-95) as one graph: 18 dependent gates on 4 x 16 bits."""
    import torch
    import nufhe_amd as nufhe
    from nufhe_amd.operators_integer import uint_min, uintarray_to_bitarray, bitarray_to_uintarray
    e = side_stream_env
    ctx, secret, cloud, stream, thr = e['ctx'], e['secret'], e['cloud'], e['stream'], e['thr']
    rs = numpy.random.RandomState(2)
    with torch.cuda.stream(stream):
        xs = [rs.randint(0, 2**16, size=4).astype(numpy.uint16) for _ in range(4)]
        a = ctx.encrypt(secret, uintarray_to_bitarray(xs[0]))
        b = ctx.encrypt(secret, uintarray_to_bitarray(xs[1]))
        answer = nufhe.api_low_level.empty_ciphertext(thr, cloud.params, (4, 16))
        g = nufhe.GateGraph(thr)
        g.capture(lambda: uint_min(thr, cloud, answer, a, b))
        g.replay()
        assert (bitarray_to_uintarray(ctx.decrypt(secret, answer)) == numpy.minimum(xs[0], xs[1])).all()
        a[...] = ctx.encrypt(secret, uintarray_to_bitarray(xs[2]))
        b[...] = ctx.encrypt(secret, uintarray_to_bitarray(xs[3]))
        g.replay()
        replayed = answer.copy()
        assert (bitarray_to_uintarray(ctx.decrypt(secret, replayed)) == numpy.minimum(xs[2], xs[3])).all()
        uint_min(thr, cloud, answer, a, b)
        assert answer == replayed


def test_captured_chain_on_the_exact_engine(side_stream_env):
    """The same kind of circuit with the NTT key on the exact-FFT engine (the split key image and the parking buffer exist
    after the eager run): the capture holds the quad kernel (3 bits) / two launches of the quad kernel for the NAND and the
    one-wave kernel with its parked accumulators for the MUX's 1200 rotations (600 bits) and replays to the eager words, also for
    new inputs written in place."""
    import torch
    import nufhe_amd as nufhe
    e = side_stream_env
    ctx, secret, stream = e['ctx'], e['secret'], e['stream']
    rs = numpy.random.RandomState(7)
    with torch.cuda.stream(stream):
        secret2, cloud = ctx.make_key_pair()
        cloud.set_engine('exact-fft')
        vm = ctx.make_virtual_machine(cloud)
        for B in (3, 600):
            ms = [rs.randint(0, 2, B).astype(bool) for _ in range(3)]
            cs = [ctx.encrypt(secret2, m) for m in ms]

            def circuit():
                t = vm.gate_nand(cs[0], cs[1])
                return vm.gate_mux(t, cs[2], cs[0])
            eager = circuit()                                   # (builds the image / sizes the parking buffer)
            g = nufhe.GateGraph(e['thr'])
            out = g.capture(circuit)
            g.replay()
            assert out == eager
            ms2 = [rs.randint(0, 2, B).astype(bool) for _ in range(3)]
            for c, m in zip(cs, ms2):
                c[...] = ctx.encrypt(secret2, m)
            expect = circuit()
            g.replay()
            assert out == expect
            assert (ctx.decrypt(secret2, out) == numpy.where(~(ms2[0] & ms2[1]), ms2[2], ms2[0])).all()
            g.close()
