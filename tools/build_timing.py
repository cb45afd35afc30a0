import time, numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # HMX_BUILD_TIMING=1: per-phase times of hmx_hmatrix_compress on stderr
import htool_amd as hm, ctypes as C
from htool_amd import _lib
from htool_amd._lib import lib, check
n = 1000000
x = hm.create_geometry("ellipse", n)
b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(100)
t = time.time(); T = b.create_cluster_tree(n, 3, x, 2, 2); print("cluster tree %.2f" % (time.time() - t))
for rep in range(2):
    tb = hm.HMatrixTreeBuilder(1e-4, 10.0, "N", "N"); tb.set_low_rank_generator("partialACA")
    tb.set_minimal_target_depth(5); tb.set_minimal_source_depth(5)
    t = time.time(); bt = tb._block_tree(T, T, -1, -1); t1 = time.time() - t
    h = C.c_void_p()
    t = time.time(); check(lib().hmx_hmatrix_create(bt, 0, C.byref(h))); t2 = time.time() - t
    params = np.array([1e-5, 1.0], dtype=np.float64)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    t = time.time(); check(lib().hmx_hmatrix_set_kernel(h, 0, dp(params), 2, 3, dp(x), dp(x))); t3 = time.time() - t
    t = time.time(); check(lib().hmx_hmatrix_compress(h, 0, 1e-4, -1)); t4 = time.time() - t
    s = _lib.Stats(); check(lib().hmx_hmatrix_stats_sized(h, C.byref(s), C.sizeof(s)))
    print("rep %d: block tree %.2f  create %.2f  set_kernel %.2f  compress %.2f (aca kernel %.3f, pack %.3f of which kernels %.3f)" % (rep, t1, t2, t3, t4, s.t_compress_s, s.t_pack_s, s.t_assemble_s))
    lib().hmx_hmatrix_destroy(h)
