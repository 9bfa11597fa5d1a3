"""Stand-in for the third-party ``numba`` package (absent from this image).

Test infrastructure only: lets the *reference* katsdpimager host path be
imported in the build container so that golden vectors can be generated
(tools/gen_golden.py).  Decorators are identities, so the reference's loops
run as interpreted Python on numpy scalars -- same arithmetic, just slow.
Never imported by the product package or on the GPU box.
"""
import numpy as np


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(fn):
        return fn
    return wrap


jit = _identity_decorator
njit = _identity_decorator


class _Type:
    def __init__(self, dtype):
        self.dtype = np.dtype(dtype)

    def __call__(self, *args):
        return (self, args)     # signature object, only inspected by vectorize below


float32 = _Type(np.float32)
float64 = _Type(np.float64)
complex64 = _Type(np.complex64)
complex128 = _Type(np.complex128)
int32 = _Type(np.int32)
int64 = _Type(np.int64)


def vectorize(signatures=None, **kwargs):
    """Element-wise ufunc emulation that keeps the declared in->out dtypes."""
    table = {}
    for ret, args in (signatures or []):
        table[args[0].dtype] = ret.dtype

    def wrap(fn):
        def call(x):
            x = np.asarray(x)
            out_dtype = table.get(x.dtype)
            if out_dtype is None:       # follow numpy casting to the widest declared input
                x = x.astype(np.float64)
                out_dtype = table.get(x.dtype, np.complex128)
            flat = x.reshape(-1)
            out = np.empty(flat.shape, out_dtype)
            typ = x.dtype.type
            for i in range(flat.size):
                out[i] = fn(typ(flat[i]))
            return out.reshape(x.shape)
        return call
    return wrap
