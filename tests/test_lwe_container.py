"""
LweSampleArray container behaviour (reference: test/test_lwe.py:397-511 -- copy, roll,
concatenate, slice assignment).  The container logic is device independent (it only manipulates the
three arrays), so these run on CPU tensors; the kernels behind the arithmetic are covered by the
GPU tests.
"""

import numpy
import pytest
import torch

from nufhe_amd.lwe import LweSampleArray, LweParams, concatenate


class _HostThread:
    """Just enough of DeviceThread for LweSampleArray.load()."""

    def to_device(self, arr):
        return torch.from_numpy(numpy.ascontiguousarray(arr))


PARAMS = LweParams(500, 0., 1.)


def mock_ciphertext(shape, seed=0):
    rs = numpy.random.RandomState(seed)
    a = rs.randint(-2**31, 2**31, size=tuple(shape) + (500,), dtype=numpy.int32)
    b = rs.randint(-2**31, 2**31, size=tuple(shape), dtype=numpy.int32)
    cv = rs.uniform(0, 1, size=tuple(shape)).astype(numpy.float32)
    return LweSampleArray(PARAMS, torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(cv))


def arrays(ct):
    return ct.a.numpy(), ct.b.numpy(), ct.current_variances.numpy()


def test_copy_is_deep_and_equal():
    ct = mock_ciphertext((3, 4, 5))
    cp = ct.copy()
    assert ct == cp
    assert cp.a.data_ptr() != ct.a.data_ptr() and cp.b.data_ptr() != ct.b.data_ptr()
    cp.b[0, 0, 0] += 1
    assert ct != cp
    assert ct != mock_ciphertext((3, 4, 5), seed=1)


@pytest.mark.parametrize('shift', [7, -9, 0])
@pytest.mark.parametrize('axis', [0, 1, -1])
def test_roll(shift, axis):
    ct = mock_ciphertext((3, 4, 5))
    rolled = ct.copy()
    rolled.roll(shift, axis=axis)
    ax = axis % 3
    for src, res in zip(arrays(ct), arrays(rolled)):
        assert (numpy.roll(src, shift, ax) == res).all()


@pytest.mark.parametrize('axis', [0, 1])
@pytest.mark.parametrize('out_none', [False, True])
def test_concatenate(axis, out_none):
    shapes = [(3, 4), (1, 4), (4, 4)] if axis == 0 else [(4, 3), (4, 1), (4, 4)]
    cts = [mock_ciphertext(s, seed=i) for i, s in enumerate(shapes)]
    out = None if out_none else mock_ciphertext((8, 4) if axis == 0 else (4, 8), seed=9)
    res = concatenate(cts, axis=axis, out=out)
    if not out_none:
        assert res is out
    for k in range(3):
        ref = numpy.concatenate([arrays(c)[k] for c in cts], axis=axis)
        assert (arrays(res)[k] == ref).all()
    with pytest.raises(ValueError):
        concatenate([])


SL = type('S', (), {'__getitem__': lambda self, i: i})()


@pytest.mark.parametrize('src_shape, src_slice, dst_shape, dst_slice', [
    ((3, 4), SL[1:], (3, 4), SL[:-1]),            # contiguous = contiguous
    ((10,), SL[1:10:2], (10,), SL[:10:2]),        # strided = strided
    ((5,), SL[1], (5,), SL[2]),                   # scalar = scalar
])
def test_slice_assignment(src_shape, src_slice, dst_shape, dst_slice):
    src = mock_ciphertext(src_shape, seed=1)
    dst = mock_ciphertext(dst_shape, seed=2)
    ref = [x.copy() for x in arrays(dst)]
    dst[dst_slice] = src[src_slice]
    for k in range(3):
        ref[k][dst_slice] = arrays(src)[k][src_slice]
        assert (arrays(dst)[k] == ref[k]).all()
    with pytest.raises(ValueError):
        dst[dst_slice] = 1


def test_shape_checks_and_serialization_roundtrip():
    with pytest.raises(ValueError):
        LweSampleArray(PARAMS, torch.zeros(3, 4, 500, dtype=torch.int32), torch.zeros(3, dtype=torch.int32),
                       torch.zeros(3, 4))
    ct = mock_ciphertext((2, 3))
    assert ct.shape == (2, 3)
    back = LweSampleArray.loads(ct.dumps(), _HostThread())
    assert back == ct and back.params == PARAMS
