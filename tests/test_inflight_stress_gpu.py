"""GPU tier: the benchmarked mode under stress -- random scenes of different sizes evaluated by 4 host threads on their own HIP
streams, 6 rounds in shuffled order, bf16 slabs (what `bench.py --inflight 4` does): every in-flight result must equal the
stand-alone result bit for bit.  Runs scripts/fuzz_inflight.py in its own process (it sets GPU_MAX_HW_QUEUES before the HIP
runtime starts, as bench.py does)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_four_streams_six_rounds_bit_identical():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_inflight.py"), "6", "4", "6"], capture_output=True,
                       text=True, timeout=900)
    print(p.stdout[-2000:])
    print(p.stderr[-2000:])
    assert p.returncode == 0, "in-flight results differ from the stand-alone ones (or a worker raised)"
    assert "mismatches: 0, errors: 0" in p.stdout
