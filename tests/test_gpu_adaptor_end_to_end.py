"""The drop-in claim end to end on the GPU: oracle/_ref/adaptor_device_check is the REAL reference (htool headers + MKL)
compiled in the dev container together with htool_amd/include/hmx/htool_adaptor.hpp and linked to libhmx.so
(oracle/ref/adaptor_device_check.cpp, `make -C oracle ref`).  It builds the same operator with htool alone on the CPU and
through every plug-in route (device kernel, htool's builder fed by the device generators, host-callback generator, uploaded
htool leaves, the distributed local operator, complex Hermitian) and compares the products.  Skipped where the binary did
not travel (it cannot be built on the GPU box: the reference tree exists only in the dev container)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "adaptor_device_check")


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/adaptor_device_check not built (needs the reference tree: dev container only)")
def test_reference_with_adaptor_on_device():
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "libs") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=900, env=env)
    print(out.stdout)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "all ok" in out.stdout and out.stdout.count(" ok\n") >= 13


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/adaptor_device_check not built (needs the reference tree: dev container only)")
def test_drop_in_routes_timed_next_to_htools_own_build():
    """`adaptor_device_check --time N`: htool's own OpenMP build, then the same operator through the device kernel, through htool's
    builder fed by the device generators (bulk block download, called concurrently from htool's OpenMP loop) and through the user's
    VirtualGenerator on all cores -- the times are printed (collected at N = 1e6 into profiles/ by tools/collect_profiles.sh), the
    results checked: ranks of the host-generator build equal the device-kernel build's leaf by leaf, products equal htool's."""
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "libs") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([EXE, "--time", "60000"], capture_output=True, text=True, timeout=900, env=env)
    print(out.stdout)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "adaptor timing: all ok" in out.stdout and out.stdout.count(" ok\n") >= 3


MPI_EXE = os.path.join(ROOT, "oracle", "_ref", "distributed_device_check")
MPIEXEC = "/opt/conda/bin/mpiexec"


@pytest.mark.skipif(not (os.path.exists(MPI_EXE) and os.path.exists(MPIEXEC)), reason="oracle/_ref/distributed_device_check or MPICH not available")
@pytest.mark.parametrize("np_", [1, 2])
def test_reference_distributed_operator_with_device_local_operator(np_):
    """examples/use_distributed_operator.cpp with htool's own MPI DistributedOperator and the rank's block rows on the GPU
    (CustomApproximationBuilder + hmx_htool::GlobalToLocalHmx), against htool's DefaultApproximationBuilder on the CPU:
    global-to-global and local-to-local products, trans N/T, symmetry N and S/U.  All ranks share the box's single GPU."""
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "libs") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([MPIEXEC, "-n", str(np_), MPI_EXE], capture_output=True, text=True, timeout=900, env=env)
    print(out.stdout)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "all ok" in out.stdout and out.stdout.count(" ok\n") >= 20
