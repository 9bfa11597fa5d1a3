"""MI355X-native imaging hot path with the operator surface of ska-sa/katsdpimager.

Sub-modules mirror the reference's: :mod:`.grid`, :mod:`.predict`, :mod:`.image`,
:mod:`.weight`, :mod:`.clean`, :mod:`.imaging`, :mod:`.parameters`.  All compute
runs in ``libkimg.so`` (hand-written HIP for gfx950, C ABI in ``include/kimg.h``);
there is no CPU fallback -- importing an operator without the built library fails.
"""
__version__ = '0.1.0'
