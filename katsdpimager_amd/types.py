"""dtype helpers (reference: katsdpimager/types.py:26-44)."""
import numpy as np


def real_to_complex(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return np.dtype(np.complex64)
    if dtype == np.float64:
        return np.dtype(np.complex128)
    raise ValueError('Unrecognised dtype {}'.format(dtype))


def complex_to_real(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.complex64:
        return np.dtype(np.float32)
    if dtype == np.complex128:
        return np.dtype(np.float64)
    raise ValueError('Unrecognised dtype {}'.format(dtype))


def require_float32(dtype, what):
    """The HIP path computes in float32/complex64 only (the reference default,
    doc/user.rst:262-266); float64 is rejected loudly rather than emulated."""
    if np.dtype(dtype) != np.float32:
        raise ValueError('{}: only float32 is supported by the HIP path, not {}'.format(
            what, np.dtype(dtype)))
