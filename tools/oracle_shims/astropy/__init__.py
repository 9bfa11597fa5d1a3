"""Stand-in for ``astropy`` (absent here): only ``astropy.units`` as a name.

The oracle bypasses the reference's parameters.py (which needs real
Quantities) by passing SimpleNamespace objects with plain floats.
"""
