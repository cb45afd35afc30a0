"""hmx_device_reserve: device arrays of 1 MiB and more are carved out of a slab taken from the driver once (first fit, neighbours
coalesced on release), hipMalloc is only asked for what does not fit.  Run in a child process: the slab belongs to the process."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import os, sys
    import numpy as np
    os.environ["HMX_NO_TORCH"] = "1"
    sys.path.insert(0, %r)
    sys.path.insert(0, os.path.join(%r, "tests"))
    import htool_amd as hm
    from helpers import load, MANIFEST
    from htool_amd._lib import lib, check
    L = lib()
    check(L.hmx_device_init(0))
    check(L.hmx_device_reserve(0, 3 << 30))

    def build(n, sym="N"):
        x = hm.create_geometry("ball", n)
        b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(50)
        T = b.create_cluster_tree(n, 3, x, 2, 2)
        tb = hm.HMatrixTreeBuilder(1e-6, 10.0, sym, "L" if sym != "N" else "N")
        tb.set_low_rank_generator("partialACA" if sym == "N" else "sympartialACA")
        return x, T, tb.build(hm.InvDistGenerator(3, x, x, 1e-5, 1.0), T, T)

    def check_product(x, T, H, n):
        rng = np.random.default_rng(n)
        u = rng.standard_normal(n)
        y = np.zeros(n)
        hm.add_hmatrix_vector_product("N", 1.0, H, u, 0.0, y)
        rows = rng.choice(n, 16, replace=False)
        exact = np.array([(1.0 / (1e-5 + np.sqrt(((x[i][None, :] - x) ** 2).sum(-1)))) @ u for i in rows])
        assert np.linalg.norm(y[rows] - exact) / np.linalg.norm(exact) < 1e-5

    # several operators alive at once, released out of order, rebuilt: ranges are split, coalesced and reused
    m0 = L.hmx_device_malloc_seconds()
    ops = [build(n, s) for n, s in ((20000, "N"), (30000, "S"), (12000, "N"))]
    for (x, T, H), n in zip(ops, (20000, 30000, 12000)):
        check_product(x, T, H, n)
    del ops[1]
    ops.append(build(25000, "N"))
    check_product(*ops[-1], 25000)
    del ops[0]
    ops.append(build(40000, "N"))  # the multi-RHS view and the transposed layout allocate more, later
    x, T, H = ops[-1]
    X = np.random.default_rng(1).standard_normal((40000, 4))
    Y = np.zeros((40000, 4))
    hm.internal_add_hmatrix_matrix_product_row_major("T", 1.0, H, X, 0.0, Y, 4)
    y1 = np.zeros(40000)
    hm.internal_add_hmatrix_vector_product("T", 1.0, H, np.ascontiguousarray(X[:, 1]), 0.0, y1)
    assert np.linalg.norm(Y[:, 1] - y1) / np.linalg.norm(y1) < 1e-12
    check_product(x, T, H, 40000)
    # an operator larger than what is left of the slab: the rest comes from hipMalloc as before
    big = build(120000, "N")
    check_product(*big, 120000)
    del ops, big, H, T, x
    import gc
    gc.collect()
    check(L.hmx_device_trim_cache())  # idle slabs go back to the driver; a second reserve works
    check(L.hmx_device_reserve(0, 1 << 30))
    check_product(*build(15000, "N"), 15000)
    assert L.hmx_device_reserve(0, 0) != 0
    print("ok")
''') % (ROOT, ROOT)


def test_device_slab_allocations():
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=900, env=dict(os.environ, HMX_NO_TORCH="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


BUDGET_SCRIPT = textwrap.dedent('''
    import ctypes as C, os, sys
    import numpy as np
    os.environ["HMX_NO_TORCH"] = "1"
    sys.path.insert(0, %r)
    import htool_amd as hm
    from htool_amd._lib import lib, check
    L = lib()
    check(L.hmx_device_init(0))
    hip = C.CDLL("libamdhip64.so")
    fr, tot = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(fr), C.byref(tot)) == 0
    check(L.hmx_device_reserve(0, fr.value - (3 << 30)))  # the driver keeps 3 GB: the cross pool's budget must come from the slab
    n = 60000
    x = hm.create_geometry("ball", n)
    b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(50)
    T = b.create_cluster_tree(n, 3, x, 2, 2)
    tb = hm.HMatrixTreeBuilder(1e-8, 10.0, "N", "N"); tb.set_low_rank_generator("partialACA")
    H = tb.build(hm.InvDistGenerator(3, x, x, 1e-5, 1.0), T, T)
    assert H.stats()["stream_bytes"] > 2.5e9  # more than the driver has left
    u = np.random.default_rng(0).standard_normal(n)
    y = np.zeros(n)
    hm.add_hmatrix_vector_product("N", 1.0, H, u, 0.0, y)
    rows = np.arange(0, n, n // 16)
    exact = np.array([(1.0 / (1e-5 + np.sqrt(((x[i][None, :] - x) ** 2).sum(-1)))) @ u for i in rows])
    assert np.linalg.norm(y[rows] - exact) / np.linalg.norm(exact) < 1e-7
    yt = np.zeros(n)
    hm.internal_add_hmatrix_vector_product("T", 1.0, H, u, 0.0, yt)  # the transposed layout is sized from the same free-memory figure
    print("ok")
''') % ROOT


def test_memory_budgets_count_the_slab():
    """With nearly all of HBM in a reserved slab the driver reports almost nothing free: the library's budgets (cross pool, views) must be
    computed from driver + slab (a 185 GB slab once made the N=1e6 ball fail with "compression pool exhausted")."""
    out = subprocess.run([sys.executable, "-c", BUDGET_SCRIPT], capture_output=True, text=True, timeout=900, env=dict(os.environ, HMX_NO_TORCH="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
