class FftTemplate:
    pass


class FftMode:
    FORWARD = 0
    INVERSE = 1
