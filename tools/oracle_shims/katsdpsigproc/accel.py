def divup(x, y):
    return (x + y - 1) // y


def roundup(x, y):
    return divup(x, y) * y


class Operation:
    pass


class OperationSequence:
    pass


class IOSlot:
    pass


class Dimension:
    pass


class DeviceArray:
    pass


class HostArray:
    pass


class DeviceAllocator:
    pass


def build(*args, **kwargs):
    raise RuntimeError('device path is not available in the oracle shim')
