"""examples/use_hmatrix.c -- the reference's examples/use_hmatrix.cpp written against the C ABI (include/hmx.h) -- built by
__graft_entry__.build() (plain gcc, linked to libhmx.so) and run on the GPU: the user's generator class as a host callback, symmetric storage,
eta = 200, epsilon = 0.01, the product in user numbering against the dense product; then the same operator from the built-in device kernel,
whose product must be bit-identical."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "use_hmatrix")


def ensure_built():
    if not os.path.exists(EXE):  # normally built by __graft_entry__.build()
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "use_hmatrix.c"), "-o", EXE,
                               "-L", os.path.join(ROOT, "htool_amd"), "-lhmx", "-Wl,-rpath,$ORIGIN/../htool_amd", "-lm"])


@pytest.mark.parametrize("n", [10000, 3000])
def test_reference_example_against_the_c_abi(n):
    ensure_built()
    out = subprocess.run([EXE, str(n)], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr[-2000:]
    err = float(re.search(r"relative error on matrix vector product : (\S+)", out.stdout).group(1))
    assert err < 0.01  # epsilon of the example (the reference prints about 1e-3 here)
    assert "bit-identical" in out.stdout
