"""C++ adaptor (htool_amd/include/hmx/htool_adaptor.hpp) against the REAL htool headers: compiles as C++14 and,
on the CPU, mirrors an htool-built H-matrix leaf for leaf.  Dev container only (the reference tree and MKL do not
exist on the GPU box): skipped elsewhere."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HTOOL = "/root/reference/include"
CONDA = "/opt/conda/lib"

pytestmark = pytest.mark.skipif(not os.path.isdir(HTOOL + "/htool") or not os.path.exists(CONDA + "/libmkl_core.so"),
                                reason="reference headers / MKL not available here")


def test_adaptor_compiles_and_mirrors_htool_structure(tmp_path):
    exe = str(tmp_path / "adaptor_host_check")
    libs = [CONDA + "/libmkl_gf_lp64.so", CONDA + "/libmkl_sequential.so", CONDA + "/libmkl_core.so"]
    cmd = ["g++", "-std=c++14", "-O1", "-Wno-deprecated-declarations", "-I" + HTOOL, "-I" + os.path.join(ROOT, "htool_amd", "include"),
           os.path.join(ROOT, "tests", "adaptor", "adaptor_host_check.cpp"), "-o", exe,
           os.path.join(ROOT, "htool_amd", "libhmx.so"), "-Wl,-rpath," + os.path.join(ROOT, "htool_amd"), "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,-rpath-link," + CONDA, "-L/opt/rocm/lib"] + libs + ["-lpthread", "-lm", "-ldl"]
    subprocess.check_call(cmd)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "libs"))
    out = subprocess.run([exe], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "missing=0" in out.stdout
