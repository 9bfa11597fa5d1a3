class AbstractContext:
    pass


class AbstractCommandQueue:
    pass


class AbstractEvent:
    pass
