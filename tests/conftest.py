import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    import torch
    from pbnet_amd.hostinfo import usable_cores
    torch.set_num_threads(usable_cores())  # the GPU boxes cap the container far below os.cpu_count()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _hang_watchdog(request):
    """A GPU test that hangs inside a native call cannot be interrupted by pytest-timeout (signals are not delivered while
    the interpreter sits in C): a watchdog thread dumps every Python stack and ends the process instead, so a hang costs
    minutes of GPU budget, not the whole call, and says where it is."""
    import faulthandler
    if request.node.get_closest_marker("gpu") is not None:
        log = open(os.path.join(ROOT, "gpurun_out", "hang_watchdog.log") if os.path.isdir(os.path.join(ROOT, "gpurun_out"))
                   else os.devnull, "a")
        log.write("== %s\n" % request.node.nodeid)
        log.flush()
        faulthandler.dump_traceback_later(int(os.environ.get("PBN_TEST_WATCHDOG", "240")), exit=True, file=log)       # pytest captures stderr: write to a file instead
        yield
        faulthandler.cancel_dump_traceback_later()
        log.close()
    else:
        yield
