class _Unit:
    def __rmul__(self, other):
        return other

    def __mul__(self, other):
        return other


m = _Unit()
Hz = _Unit()
rad = _Unit()


def Quantity(x, *args, **kwargs):
    return x


def spectral():
    return None
