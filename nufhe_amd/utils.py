import numpy


def to_numpy(arr):
    """Host copy of a device (torch) or host array."""
    if hasattr(arr, 'detach'):
        return arr.detach().cpu().numpy()
    return numpy.asarray(arr)


def arrays_equal(arr1, arr2):
    """nufhe/utils.py:18-20"""
    a1 = to_numpy(arr1)
    a2 = to_numpy(arr2)
    return a1.shape == a2.shape and bool((a1 == a2).all())
