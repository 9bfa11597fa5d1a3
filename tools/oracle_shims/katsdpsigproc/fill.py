class FillTemplate:
    pass
