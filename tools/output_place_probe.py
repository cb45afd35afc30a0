#!/usr/bin/env python3
"""Runs ON THE GPU BOX: ONE operator (N = 1e6 fp64, the headline configuration), its products timed per kernel with the OUTPUT vector at
given places of the reserved slab (hmx_device_slab_alloc_at), then where hmx_hmatrix_alloc_vector puts it and where torch.zeros does: do the
two speeds of the expand kernels follow the place of the vector they write?"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Raw:
    def __init__(self, ptr, shape, dt):
        self.__cuda_array_interface__ = dict(shape=shape, typestr=np.dtype(dt).str, data=(ptr, False), version=2, strides=None)


def main():
    import torch
    import bench
    import htool_amd as hm
    from htool_amd._lib import check
    n, mu = 1000000, 16
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    x = hm.create_geometry("ellipse", n)
    ctb = hm.ClusterTreeBuilder()
    ctb.set_maximal_leaf_size(100)
    T = ctb.create_cluster_tree(n, 3, x, 2, 2)
    L = hm.lib()
    L.hmx_device_init(0)
    free_b, _ = torch.cuda.mem_get_info(0)
    L.hmx_device_reserve(0, int(0.6 * free_b))
    gen = hm.InvDistGenerator(3, x, x, 1e-5, 1.0, 1.0, 0.0, False)
    tb = hm.HMatrixTreeBuilder(1e-4, 10.0, "N", "N")
    tb.set_low_rank_generator("partialACA")
    d = bench.minimal_depth(n)
    tb.set_minimal_target_depth(d)
    tb.set_minimal_source_depth(d)
    tb.set_option("build_timing", 1)
    H = tb.build(gen, T, T, -1, -1, device=0, dtype=np.float64)
    rng = np.random.default_rng(1)
    X = torch.from_numpy(rng.random((n, mu))).to(dev)
    x1 = torch.from_numpy(rng.random(n)).to(dev)

    def timed(Y, y1):
        out = {}
        for name, fn in (("mu16", lambda: hm.internal_add_hmatrix_matrix_product_row_major("N", 1.0, H, X, 0.0, Y, mu)),
                         ("one", lambda: hm.internal_add_hmatrix_vector_product("N", 1.0, H, x1, 0.0, y1))):
            for _ in range(3):
                fn()
            H.set_profiling(True)
            acc = {}
            for _ in range(5):
                fn()
                for k, t in H.last_kernel_times():
                    acc.setdefault(k, []).append(t)
            H.set_profiling(False)
            out[name] = {k: round(float(np.mean(v)), 3) for k, v in acc.items()}
        return out

    ref = None
    nbY, nby = n * mu * 8, n * 8
    for frac in [0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 1.0]:
        pY, py = C.c_void_p(), C.c_void_p()
        check(L.hmx_device_slab_alloc_at(0, nbY, frac, C.byref(pY)))
        check(L.hmx_device_slab_alloc_at(0, nby, frac, C.byref(py)))
        Y = torch.as_tensor(Raw(pY.value, (n, mu), np.float64), device=dev)
        y1 = torch.as_tensor(Raw(py.value, (n,), np.float64), device=dev)
        t = timed(Y, y1)
        if ref is None:
            ref = (Y.clone(), y1.clone())
        same = bool(torch.equal(Y, ref[0]) and torch.equal(y1, ref[1]))
        print("[output place] slab fraction %.3f (Y at %#x): %s results bitwise equal: %s" % (frac, pY.value, json.dumps(t), same), flush=True)
        del Y, y1
        torch.cuda.synchronize()
        check(L.hmx_device_slab_free(0, pY, nbY))
        check(L.hmx_device_slab_free(0, py, nby))
    Y, y1 = H.empty_output((n, mu)), H.empty_output(n)
    t = timed(Y, y1)
    print("[output place] hmx_hmatrix_alloc_vector (Y at %#x): %s results bitwise equal: %s" % (Y.data_ptr(), json.dumps(t), bool(torch.equal(Y, ref[0]) and torch.equal(y1, ref[1]))), flush=True)
    Y, y1 = torch.zeros((n, mu), dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev)
    t = timed(Y, y1)
    print("[output place] torch.zeros (Y at %#x): %s" % (Y.data_ptr(), json.dumps(t)), flush=True)


if __name__ == "__main__":
    main()
