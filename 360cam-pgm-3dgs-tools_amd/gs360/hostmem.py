"""Host allocator settings for the codec threads.

Every decoded panorama and every encoded view is a 8-50 MB NumPy / Pillow buffer that lives for a few tens of milliseconds.  glibc
hands such blocks back to the kernel as soon as they are freed (heap trimming), so each of the dozens of decode / encode threads
page-faults its buffers in again and again, and the faults of all threads serialise on the process's memory-map lock: measured on the
MI355X box (256 host threads, `scripts/bench_cli_e2e.py --frames 48 --jobs 32`) 33 -> 44-53 frames/s with trimming off.
`tune_malloc()` tells glibc to keep freed memory (M_TRIM_THRESHOLD) and to grow heaps in larger steps (M_TOP_PAD).  Process-wide,
glibc only, idempotent; `GS360_MALLOC_TUNE=0` leaves the allocator alone.
"""
import ctypes
import os

_M_TRIM_THRESHOLD = -1
_M_TOP_PAD = -2
_done = False


def tune_malloc() -> bool:
    """-> True when the settings were applied (now or earlier)."""
    global _done
    if _done:
        return True
    if os.environ.get("GS360_MALLOC_TUNE", "1") in ("0", "off", "no"):
        return False
    try:
        libc = ctypes.CDLL("libc.so.6")
        mallopt = libc.mallopt
    except (OSError, AttributeError):
        return False
    mallopt.argtypes = [ctypes.c_int, ctypes.c_int]
    mallopt.restype = ctypes.c_int
    ok = mallopt(_M_TRIM_THRESHOLD, 2**31 - 1) == 1
    ok = (mallopt(_M_TOP_PAD, 256 << 20) == 1) and ok
    _done = ok
    return ok


def effective_cpus() -> int:
    """CPUs this process can actually use: the scheduler affinity mask, cut down by a cgroup CPU quota when there is one (cgroup v2
    `cpu.max`, v1 `cpu.cfs_quota_us` / `cpu.cfs_period_us`).  os.cpu_count() alone reports the host's 256 hardware threads inside a
    container that is allowed 16 CPUs' worth of time -- the MI355X boxes of the build pool are such containers -- and worker pools
    sized by it thrash: there the codecs saturate at ~16 threads whatever `-j` says."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            quota = None
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)
