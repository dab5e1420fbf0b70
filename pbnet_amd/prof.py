"""Per-stage wall timers (the reference has none beyond AverageMeters).  Disabled by default;
PBNET_PROF=1 or prof.enable() makes every `with section(name)` synchronise the device on both sides and accumulate."""
import contextlib
import os
import time

ENABLED = os.environ.get("PBNET_PROF", "0") == "1"
TIMES = {}
COUNTS = {}


def enable(on=True):
    global ENABLED
    ENABLED = on


def reset():
    TIMES.clear()
    COUNTS.clear()


@contextlib.contextmanager
def section(name):
    if not ENABLED:
        yield
        return
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        yield
    finally:
        torch.cuda.synchronize()
        TIMES[name] = TIMES.get(name, 0.0) + time.perf_counter() - t0
        COUNTS[name] = COUNTS.get(name, 0) + 1


def report():
    return {k: (1e3 * v / COUNTS[k], COUNTS[k]) for k, v in TIMES.items()}


# ---- host-side marks: wall-clock stamps WITHOUT device synchronisation (where does the Python thread spend its time) ----
HOST_MARKS = None


def host_marks(enable=True):
    """Start (or stop) recording; returns the list that `mark` appends (name, perf_counter seconds) to."""
    global HOST_MARKS
    HOST_MARKS = [] if enable else None
    return HOST_MARKS


def mark(name):
    if HOST_MARKS is not None:
        HOST_MARKS.append((name, time.perf_counter()))
