"""roctx ranges around the phases of a training step, so that a rocprofv3 --marker-trace / --kernel-trace timeline reads as
tower forward | exchange | loss | backward | gradient all-reduce | optimiser instead of ~500 anonymous launches.

Off unless MSN_ROCTX=1 is in the environment (the range calls cost a few hundred nanoseconds each and need libroctx64.so
from /opt/rocm/lib); `with markers.range("name"):` is a no-op otherwise.  Ranges are host-side brackets around the ENQUEUE of a
phase: the kernels they cover are the ones whose launches fall inside the bracket."""
import contextlib
import ctypes
import os

ENABLED = os.environ.get("MSN_ROCTX", "0") == "1"
_lib = None


def _roctx():
    global _lib, ENABLED
    if _lib is None:
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so",
                     "/opt/rocm/lib/librocprofiler-sdk-roctx.so"):
            try:
                _lib = ctypes.CDLL(name)
                _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                _lib.roctxRangePushA.restype = ctypes.c_int
                _lib.roctxRangePop.restype = ctypes.c_int
                break
            except (OSError, AttributeError):
                _lib = None
        if _lib is None:
            ENABLED = False
    return _lib


@contextlib.contextmanager
def _live(name):
    lib = _roctx()
    if lib is None:
        yield
        return
    lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        lib.roctxRangePop()


_NULL = contextlib.nullcontext()


def range(name):
    """Context manager: a roctx range called `name` (a shared no-op context when markers are off)."""
    return _live(name) if ENABLED else _NULL
