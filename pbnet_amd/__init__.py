"""pbnet_amd -- MI355X-native hot path of PBNet.

Importing the package caps torch's intra-op CPU thread pool at the host core budget (scheduler affinity and cgroup
CPU quota).  The host side of the path is a thin driver, but torch sizes its OpenMP pool from os.cpu_count(); on a box
that exposes 256 logical CPUs and caps the container at 16, the pool's spinning workers exhaust the cgroup quota and
the whole process -- including the thread that feeds the GPU -- is frozen for tens of milliseconds every period
(measured: every third 17 ms forward took 66 ms).  Set PBNET_KEEP_TORCH_THREADS=1 to leave the pool alone."""
import os as _os


def _cap_host_threads():
    if _os.environ.get("PBNET_KEEP_TORCH_THREADS", "0") == "1":
        return
    try:
        import torch
        from .hostinfo import usable_cores
        # one process per GPU: the ranks of a node share the budget (torchrun exports LOCAL_WORLD_SIZE)
        ranks = max(1, int(_os.environ.get("LOCAL_WORLD_SIZE", _os.environ.get("WORLD_SIZE", "1")) or 1))
        budget = max(1, min(usable_cores() // ranks, 16))
        if torch.get_num_threads() > budget:
            torch.set_num_threads(budget)
    except Exception:  # never make the import fail on an exotic host
        pass


_cap_host_threads()
