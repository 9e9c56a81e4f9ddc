"""Per-kernel HIP-event timing of the library (fd_profile_begin / fd_profile_end)."""
import ctypes as C

from . import _lib


def begin(only=None):
    """Starts bracketing every kernel group, or only the group named ``only``."""
    if only:
        _lib.check(_lib.lib().fd_profile_begin_only(only.encode("ascii")))
    else:
        _lib.check(_lib.lib().fd_profile_begin())


def end(capacity=64):
    """Stops profiling (device sync) and returns [{kernel, launches, ms, flops, bytes}, ...]."""
    rows = (_lib.ProfileRow * capacity)()
    n = C.c_int(0)
    _lib.check(_lib.lib().fd_profile_end(rows, capacity, C.byref(n)))
    out = []
    for i in range(min(n.value, capacity)):
        r = rows[i]
        out.append(dict(kernel=r.kernel.decode("ascii"), launches=int(r.launches), ms=float(r.ms),
                        flops=float(r.flops), bytes=float(r.bytes)))
    return out
