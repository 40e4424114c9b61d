"""Host build (CPU emulator) of the redundant-limb arithmetic and of the limb-form transforms of the
blind-rotation loop (nufhe_amd/csrc/ff24.h, ntt1024_l4.h) against Python integers and against the
64-bit transforms of ntt1024.h (which are pinned to the reference's ntt_transform_ref by
tests/test_emu_device_code.py)."""
import numpy

from tests import l4_checks
from tests.emu import emu

P = l4_checks.P


def test_l4_primitives_vs_bigint():
    l4_checks.check_all(lambda op, a, b=None, c=None, shift=0: emu.l4_op(op, a, b, c, shift))


def test_forward_small_l4_equals_u64_transform():
    rs = numpy.random.RandomState(11)
    for trial in range(4):
        d = rs.randint(-512, 512, size=1024).astype(numpy.int32)
        if trial == 1:
            d[:] = -512
        if trial == 2:
            d[:] = 511
            d[::3] = -512
        ref = emu.ntt_forward(numpy.array([int(x) % P for x in d], dtype=numpy.uint64))
        got = emu.ntt_forward_small_l4(d)
        assert all(int(g) % P == int(r) for g, r in zip(got, ref))


def test_inverse_l4_equals_u64_transform_and_round_trip():
    rs = numpy.random.RandomState(12)
    # coefficients of magnitude < 2^52 (the range of an external product)
    c = rs.randint(-2**52, 2**52, size=1024, dtype=numpy.int64)
    c[:4] = [2**52 - 1, -2**52 + 1, 0, -1]
    spec = emu.ntt_forward(numpy.array([int(x) % P for x in c], dtype=numpy.uint64))
    # any 64-bit representative is accepted: add P where it fits
    rep = numpy.array([int(s) + P if int(s) + P < 2**64 and i % 2 else int(s) for i, s in enumerate(spec)], dtype=numpy.uint64)
    got = emu.ntt_inverse_l4_i32(rep)
    assert (got == (c & 0xFFFFFFFF).astype(numpy.uint32)).all()
    # digits -> forward (limbs) -> inverse (limbs) -> digits
    d = rs.randint(-512, 512, size=1024).astype(numpy.int32)
    back = emu.ntt_inverse_l4_i32(emu.ntt_forward_small_l4(d))
    assert (back.view(numpy.int32) == d).all()
