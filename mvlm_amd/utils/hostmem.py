"""Host allocator hint for scan ingest.

Every scan brings 10-30 MB of fresh host arrays (parsed geometry, decoded texture) and drops them a few milliseconds
later.  glibc serves allocations of that size with mmap and returns them with munmap - and in a process that has the
GPU open, mapping and unmapping pages is slow (the driver's MMU notifier runs on every range): measured on the MI355X
box, 10 ms of page faults while a scan is parsed and 6.5 ms to drop it, a third of an 8-view ``predict_one_file``.
``retain_freed_host_memory()`` tells glibc to serve blocks up to 32 MB (its maximum) from the heap and to keep up to
1 GB of freed heap instead of trimming it, so the next scan reuses the previous one's pages (47.9 -> 30.7 ms per 8-view
scan, ingest included).  Process-wide, glibc only, applied once by ``Pipeline``; ``MVLM_HOST_MALLOC_TUNING=0`` opts out.
"""
from __future__ import annotations

import ctypes
import os

_M_TRIM_THRESHOLD = -1
_M_MMAP_THRESHOLD = -3
_applied: bool | None = None


def retain_freed_host_memory() -> bool:
    """Apply the hint once; returns whether it is in effect (False: opted out, or not glibc)."""
    global _applied
    if _applied is not None:
        return _applied
    _applied = False
    if os.environ.get("MVLM_HOST_MALLOC_TUNING", "1") == "0":
        return False
    try:
        libc = ctypes.CDLL("libc.so.6")
        mallopt = libc.mallopt
    except (OSError, AttributeError):
        return False
    mallopt.argtypes = [ctypes.c_int, ctypes.c_int]
    mallopt.restype = ctypes.c_int
    ok = mallopt(_M_MMAP_THRESHOLD, 32 << 20) == 1 and mallopt(_M_TRIM_THRESHOLD, 1 << 30) == 1
    _applied = bool(ok)
    return _applied
