"""Host allocator settings for the codec threads of the drop-in CLIs.

Every decoded panorama and every encoded view is a 8-50 MB NumPy / Pillow buffer that lives for a few tens of milliseconds, in dozens
of decode / encode threads at once.  `tune_malloc()` makes three glibc settings (mallopt): heap trimming off (M_TRIM_THRESHOLD), heaps
grown in 256 MB steps (M_TOP_PAD), and the mmap threshold raised to its maximum of 32 MiB (M_MMAP_THRESHOLD) so that blocks up to
that size are served from the heaps and recycled instead of being mapped and unmapped once per image (any mallopt call freezes glibc's
dynamic threshold; left at its 128 KiB start value every one of these buffers would be an mmap / munmap pair and the first two
settings would not touch them).  Blocks above 32 MiB (a decoded 5.7K panorama is 50 MB) stay mmap'ed.  Measured on the MI355X box
(256 host threads under a 16-CPU quota, `scripts/bench_cli_e2e.py --frames 48 --jobs 32`): profiles/r03/cli_e2e_malloc.txt (first
two settings: 34-36 -> 40-47 frames/s) and profiles/r04/cli_e2e_malloc_mmap.txt (all three: 45-47 frames/s with or without the third --
the run is bound by the codecs under the CPU quota by then).

Process-wide and permanent (a process that has called it never trims again and keeps a 256 MB top pad per heap), so it is the
CLIs' `main()` that calls it, not the engine: a host application that imports the engine keeps its own allocator behaviour.
glibc only, idempotent; `GS360_MALLOC_TUNE=0` leaves the allocator alone, `GS360_MALLOC_MMAP_MB=0` leaves the mmap threshold alone.
"""
import ctypes
import os

_M_TRIM_THRESHOLD = -1
_M_TOP_PAD = -2
_M_MMAP_THRESHOLD = -3
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
    try:
        mmap_mb = int(os.environ.get("GS360_MALLOC_MMAP_MB", "32"))
    except ValueError:
        mmap_mb = 32
    if mmap_mb > 0:
        ok = (mallopt(_M_MMAP_THRESHOLD, min(mmap_mb, 32) << 20) == 1) and ok
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
