"""Stand-in for ``astropy.modeling`` (absent here): names only.  The beam goldens pass a
SimpleNamespace with the attributes of a fitted Gaussian2D instead of a real model."""
from types import SimpleNamespace

models = SimpleNamespace()
fitting = SimpleNamespace()
