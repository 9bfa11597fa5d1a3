"""Named ranges for rocprofv3's marker trace (``rocprofv3 --marker-trace``), the counterpart of
the NVTX ranges SURVEY section 5 lists among the reference's tracing aids.

Off by default and free when off; ``trace.enable()`` loads ROCm's roctx library and
makes :func:`range` push / pop real ranges.  The stages of ``frontend.process_channel`` and the
per-chunk work of ``make_weights`` / ``make_dirty`` carry ranges; ``bench.py --roctx`` switches
them on for the major-cycle loop."""
import contextlib
import ctypes

_lib = None


#: rocprofv3 records the ranges of the SDK's roctx library; the older libroctx64 is what roctracer
#: based tools see
LIBRARIES = ('librocprofiler-sdk-roctx.so', 'libroctx64.so')


def enable(library=None):
    """Load roctx; returns False (and stays off) when no library is there."""
    global _lib
    if _lib is None:
        for name in ((library,) if library else LIBRARIES):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _lib = lib
                break
            except (OSError, AttributeError):
                continue
    return _lib is not None


def disable():
    global _lib
    _lib = None


def enabled():
    return _lib is not None


#: host-side timeline: when a list is installed with :func:`record_timeline`, every range also
#: appends (thread name, range name, start, end) in ``time.perf_counter`` seconds -- what the host
#: threads of ``frontend.process_channels`` were doing when, without a profiler
_timeline = None


def record_timeline(entries):
    """Install (or, with None, remove) the list that collects host-side range timings."""
    global _timeline
    _timeline = entries


@contextlib.contextmanager
def range(name):        # noqa: A001  (the name roctx and NVTX use)
    """``with trace.range('grid'):`` -- a named range on the calling thread."""
    lib = _lib
    timeline = _timeline
    if lib is None and timeline is None:
        yield
        return
    if timeline is not None:
        import threading
        import time
        t0 = time.perf_counter()
    if lib is not None:
        lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        if lib is not None:
            lib.roctxRangePop()
        if timeline is not None:
            timeline.append((threading.current_thread().name, name, t0, time.perf_counter()))
