"""The C-level DistributedOperator (hmx_dist_*, include/hmx.h) without torch: libhmx + the system RCCL, one rank (the only size one
GPU offers; with HMX_DIST_FORCE_COLLECTIVES=1 the collectives are really issued on a one-rank communicator, through both the
all-gather and the grouped-broadcast route).  Device memory through the HIP runtime directly.  Subprocess: HMX_NO_TORCH must be set
before libhmx is loaded."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import ctypes as C, os, sys
    import numpy as np
    os.environ["HMX_NO_TORCH"] = "1"
    sys.path.insert(0, %r)
    import htool_amd as hm
    from htool_amd._lib import lib, check
    L = lib()
    hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    rccl = C.CDLL("/opt/rocm/lib/librccl.so", mode=C.RTLD_GLOBAL)
    class Uid(C.Structure):
        _fields_ = [("b", C.c_char * 128)]
    uid, comm = Uid(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    def dev(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0
        return p
    def host(p, like):
        out = np.empty_like(like)
        assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), p, C.c_size_t(out.nbytes), 2) == 0
        return out
    n = 5000
    x3 = hm.create_geometry("ellipse", n)
    b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(64)
    T = b.create_cluster_tree(n, 3, x3, 2, 1)
    rng = np.random.default_rng(0)
    for dtype in (np.float64, np.complex128, np.float32):
        cplx = dtype == np.complex128
        tb = hm.HMatrixTreeBuilder(1e-5, 10.0, "N", "N"); tb.set_low_rank_generator("partialACA")
        H = tb.build(hm.InvDistGenerator(3, x3, x3, 1e-5, 1.0, 1.0, 0.5 if cplx else 0.0), T, T, 0, 0, dtype=dtype)
        D = C.c_void_p()
        check(L.hmx_dist_create(H._h, T._h, T._h, comm, 0, 1, None, C.byref(D)))
        xin = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dtype)
        y0 = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dtype)
        ab = np.array([1.5 - (0.5j if cplx else 0), 0.25 + (1j if cplx else 0)], dtype=dtype)
        pa, pb = C.c_void_p(ab.ctypes.data), C.c_void_p(ab.ctypes.data + ab.itemsize)
        for trans in ("N", "T"):
            ref = y0.copy()
            hm.internal_add_hmatrix_vector_product(trans, ab[0], H, xin, ab[1], ref)
            for fn in (L.hmx_dist_matvec_global_to_global, L.hmx_dist_matvec_local_to_local):
                dx, dy = dev(xin), dev(y0)
                check(fn(D, trans.encode(), pa, dx, pb, dy, None))
                assert hip.hipDeviceSynchronize() == 0
                y = host(dy, y0)
                err = np.linalg.norm(y - ref) / np.linalg.norm(ref)
                assert err < (1e-5 if dtype == np.float32 else 1e-13), (np.dtype(dtype).name, trans, fn.__name__, err)
                hip.hipFree(dx); hip.hipFree(dy)
        L.hmx_dist_destroy(D)
        print("ok", np.dtype(dtype).name)
''') % ROOT


@pytest.mark.parametrize("env", [{}, {"HMX_DIST_FORCE_COLLECTIVES": "1"}, {"HMX_DIST_FORCE_COLLECTIVES": "1", "HMX_DIST_NO_ALLGATHER": "1"}])
def test_c_level_distributed_operator(env):
    e = dict(os.environ, HMX_NO_TORCH="1", **env)
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("ok ") == 3, out.stdout
