"""A short run of the randomised parity sweep (tools/fuzz_parity.py): device engine vs CPU oracle on random geometry / size / leaf
size / children / partitions / strategy / eta / eps / compressor / symmetry / coefficient type / row partition / minimal depth;
structure and (double precision) ranks must be identical, single and multiple right-hand-side products within tolerance."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_configurations_against_the_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "40", "7"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "random configurations ok" in out.stdout
