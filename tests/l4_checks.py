"""Shared checks of the redundant-limb arithmetic (nufhe_amd/csrc/ff24.h, ntt1024_l4.h) against Python
integers.  `run(op, a, b=None, c=None, shift=0) -> (out, out2)` executes the test-hook dispatcher of
csrc/l4_hook.h on uint32 [n, 4] arrays: tests/test_emu_l4.py binds it to the host build (CPU emulator),
tests/test_gpu_kernels.py to the gfx950 build (nufhe_l4_op), so the device-only instruction sequences
(v_perm_b32 selectors, the carry-out of v_mad_u64_u32) are compared with the same expectations."""
import numpy

P = 2**64 - 2**32 + 1
Z = (0x7f004083, 0xbfbffc81, 0x4002ff40, 0x80000000)


def value(limbs):
    """field value of an int32 [4] limb vector"""
    def s32(w):
        w = int(w)
        return w - 2**32 if w >= 2**31 else w
    return sum(s32(w) << (24 * i) for i, w in enumerate(limbs)) % P


def as_u32(rows):
    return numpy.array([[w & 0xFFFFFFFF for w in r] for r in rows], dtype=numpy.uint32)


def signed(rows):
    return numpy.asarray(rows, numpy.uint32).view(numpy.int32).astype(object)


def u64_limbs(vals):
    """uint64 values as two u32 words in columns 0, 1"""
    return [[v & 0xFFFFFFFF, v >> 32, 0, 0] for v in vals]


# limb vectors that represent 0: the rotations of P = (1, -2^8, 2^16, 0) and the carry moves
_ZERO = [(1, -2**8, 2**16, 0), (0, 1, -2**8, 2**16), (-2**16, 0, 1, -2**8), (2**8, -2**16, 0, 1),
         (-2**24, 1, 0, 0), (0, -2**24, 1, 0), (0, 0, -2**24, 1), (1, 0, 0, 2**24)]


def spread(rs, v):
    """a non-normalised limb vector (|w| < 2^30) of the canonical value v"""
    w = [v & 0xFFFFFF, (v >> 24) & 0xFFFFFF, v >> 48, 0]
    for z, kmax in zip(_ZERO, (2**11, 2**11, 2**11, 2**11, 30, 30, 30, 30)):
        k = int(rs.randint(-kmax, kmax + 1))
        w = [x + k * y for x, y in zip(w, z)]
    assert all(abs(x) < 2**30 for x in w) and value(w) == v
    return w


def random_limbs(rs, n, bound=2**30):
    rows = rs.randint(-bound, bound + 1, size=(n, 4), dtype=numpy.int64).tolist()
    edge = [[bound] * 4, [-bound] * 4, [0] * 4, [bound, -bound, bound, -bound], [-bound, bound, -bound, bound],
            [0, 0, 0, bound], [0, 0, 0, -bound], [-1, 0, 0, 0], [1, 0, 0, 0], [0, 0, -1, -1]]
    # non-normalised representations of values around 0 and P (carry chain, top-limb fold)
    if bound >= 2**30:
        for v in (0, 1, 2, P - 1, P - 2, 2**32 - 1, 2**32, 2**32 - 2, 2**63, P // 2, P // 2 + 1, P - 2**32):
            edge += [spread(rs, v) for _ in range(4)]
    return edge + rows


def check_offsets():
    assert sum(z << (24 * i) for i, z in enumerate(Z)) % P == 0
    # u_i = w_i + Z_i stays in [0, 2^32) for |w_i| <= 2^30, including what l4_to_u64 adds on top:
    # the chain's carries (< 2^9) on limbs 1, 2 and the folded top limb (< 2^24) on limb 2; limb 1 also
    # gives up to 2^16
    assert all(z >= 2**30 + 2**17 for z in Z)
    assert Z[0] + 2**30 < 2**32 and Z[3] + 2**30 < 2**32
    assert Z[1] + 2**30 + 2**9 < 2**32 and Z[2] + 2**30 + 2**24 + 2**9 < 2**32


def check_all(run, rs=None, n=3000):
    rs = rs or numpy.random.RandomState(7)
    check_offsets()
    rows = random_limbs(rs, n)
    A = as_u32(rows)
    vals = [value(r) for r in rows]

    # 0: L4 -> 64-bit representative
    out, _ = run(0, A)
    got = [int(o[0]) | (int(o[1]) << 32) for o in out]
    assert all(g % P == v for g, v in zip(got, vals))
    assert all(int(o[2]) == 0 and int(o[3]) == 0 for o in out)

    # 6: canonical-range conversion to int32 for values known to be small integers
    small = [int(x) for x in rs.randint(-2**61, 2**61, size=500, dtype=numpy.int64)] + [0, 1, -1, 2**32 - 2, 2**32 - 1,
                                                                                      2**32, -2**32, 2**52, -2**52]
    rows6 = [spread(rs, cval % P) for cval in small]
    out, _ = run(6, as_u32(rows6))
    assert [int(o[0]) for o in out] == [cval & 0xFFFFFFFF for cval in small]

    # 10: the same with a sub-limb power of two folded in: (value * 2^s) is the small integer
    for sh in (0, 6, 12, 18):
        rows10 = [spread(rs, (cval * pow(2, 192 - sh, P)) % P) for cval in small]     # value = c * 2^-sh
        rows10 = [r for r in rows10]
        out, _ = run(10, as_u32(rows10), shift=sh)
        assert [int(o[0]) for o in out] == [cval & 0xFFFFFFFF for cval in small], sh

    # 1: 128-bit product -> limbs;  7: 64-bit word -> limbs
    words = [int(x) for x in rs.randint(0, 2**63, size=n, dtype=numpy.int64).astype(object) * 2 + rs.randint(0, 2, size=n)]
    words2 = [int(x) for x in rs.randint(0, 2**63, size=n, dtype=numpy.int64).astype(object) * 2 + 1]
    words[:4] = [0, 2**64 - 1, P, P - 1]; words2[:4] = [2**64 - 1, 2**64 - 1, 0, P - 1]
    out, _ = run(1, as_u32([[lo & 0xFFFFFFFF, lo >> 32, hi & 0xFFFFFFFF, hi >> 32] for lo, hi in zip(words, words2)]))
    so = signed(out)
    assert all(value(o) == (lo + (hi << 64)) % P for o, lo, hi in zip(out, words, words2))
    assert (abs(so[:, 0]) < 2**24).all() and (so[:, 1] > -2**8).all() and (so[:, 1] < 2**24).all()
    assert (so[:, 2] >= 0).all() and (so[:, 2] < 2**24).all() and (so[:, 3] >= 0).all() and (so[:, 3] < 2**24).all()
    out, _ = run(7, as_u32(u64_limbs(words)))
    assert all(value(o) == w % P for o, w in zip(out, words))
    assert (signed(out) >= 0).all() and (signed(out) < 2**24).all()

    # 2: every compile-time power of two, limbs up to 2^28 (the pass outputs they are applied to)
    rows2 = random_limbs(rs, 200, 2**28)
    A2 = as_u32(rows2)
    v2 = [value(r) for r in rows2]
    for shift in range(192):
        out, _ = run(2, A2, shift=shift)
        assert all(value(o) == (v << shift) % P for o, v in zip(out, v2)), shift
        if shift % 24:
            assert (abs(signed(out)) <= 2**24 + 2**(28 - 24 + shift % 24)).all(), shift

    # 3: butterflies with every twiddle 2^(6 s)
    B2 = as_u32(random_limbs(rs, 200, 2**28)[::-1])
    vb = [value(r) for r in B2.view(numpy.int32)]
    for shift in range(0, 192, 6):
        out, out2 = run(3, A2, B2, shift=shift)
        assert all(value(o) == (x + y) % P for o, x, y in zip(out, v2, vb)), shift
        assert all(value(o) == ((x - y) << shift) % P for o, x, y in zip(out2, v2, vb)), shift

    # 4: general multiplication by a 64-bit factor
    tw = words[:len(rows)] if len(words) >= len(rows) else (words * (len(rows) // len(words) + 1))[:len(rows)]
    out, _ = run(4, A, as_u32(u64_limbs(tw)))
    assert all(value(o) == (v * t) % P for o, v, t in zip(out, vals, tw))

    # 5: two products + addend straight into limbs (any 64-bit operands, maximal ones included)
    a0 = words[:n]; b0 = words2[:n]; a1 = words2[::-1][:n]; b1 = words[::-1][:n]; cc = words[5:] + words[:5]
    a0[:2] = [2**64 - 1] * 2; b0[:2] = [2**64 - 1] * 2; a1[:2] = [2**64 - 1] * 2; b1[:2] = [2**64 - 1] * 2
    cc[:2] = [2**64 - 1, 0]
    out, _ = run(5, as_u32([[x & 0xFFFFFFFF, x >> 32, y & 0xFFFFFFFF, y >> 32] for x, y in zip(a0, b0)]),
                 as_u32([[x & 0xFFFFFFFF, x >> 32, y & 0xFFFFFFFF, y >> 32] for x, y in zip(a1, b1)]),
                 as_u32(u64_limbs(cc[:n])))
    assert all(value(o) == (x * y + z * w + c) % P for o, x, y, z, w, c in zip(out, a0, b0, a1, b1, cc))
    so = signed(out)
    assert (abs(so[:, 0]) < 2**24).all() and (so[:, 1] > -2**11).all() and (so[:, 1] < 2**24).all()

    # 8: per-lane twiddles of both directions, all four lane classes
    for direction in (0, 1):
        for hi in range(4):
            for q in (1, 2, 3):
                for lo in range(4):
                    e = (12 * q * hi + 3 * q * lo) * (1 if direction == 0 else -1)
                    lo_col = as_u32([[lo, 0, 0, 0]] * len(rows2))
                    out, _ = run(8, A2, lo_col, shift=12 * direction + 3 * hi + (q - 1))
                    assert all(value(o) == (v * pow(2, e % 192, P)) % P for o, v in zip(out, v2)), (direction, hi, q, lo)
                    assert (abs(signed(out)) < 2**26).all()

    # 9: digit placement
    digits = list(range(-512, 512, 37)) + [-512, 511, 0, 1, -1]
    for j2 in range(16):
        out, _ = run(9, as_u32([[d, 0, 0, 0] for d in digits]), shift=j2)
        assert all(value(o) == (d << (6 * j2)) % P for o, d in zip(out, digits)), j2
        assert (abs(signed(out)) <= 2**27).all()
