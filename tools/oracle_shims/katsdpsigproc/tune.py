def autotuner(test=None):
    def wrap(fn):
        return fn
    return wrap


def autotune(*args, **kwargs):
    raise RuntimeError('device path is not available in the oracle shim')


def make_measure(*args, **kwargs):
    raise RuntimeError('device path is not available in the oracle shim')
