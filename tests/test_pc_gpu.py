"""GPU tier: the pair-compacted convolution family of round 6 (pbnet_amd/csrc/experiments/spconv_pc.hip) is NOT in the product
library -- it is correct and slower than what ships.  Its parity cases (tests/experiments/pc_cases.py:
every built shape against the CPU oracle and against k_spconv, bit for bit, strided / transposed maps, the folded shortcut, a
device-side row count) run in a child process against the experiments library (`make -C pbnet_amd/csrc experiments`, built by
__graft_entry__.build())."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pbnet_amd", "libpbnet_hip_exp.so")


@pytest.mark.skipif(not os.path.exists(LIB), reason="experiments library not built (make -C pbnet_amd/csrc experiments)")
def test_pair_compacted_family_in_the_experiments_library():
    env = dict(os.environ, PBNET_HIP_LIB=LIB)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "experiments", "pc_cases.py"), "-q", "-x", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(p.stdout[-3000:])
    print(p.stderr[-1500:])
    assert p.returncode == 0, "pair-compacted parity cases failed against the experiments library"
    assert " passed" in p.stdout
