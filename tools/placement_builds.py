#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the same operator built several times in ONE process (every build takes new ranges of the reserved slab; a dummy
allocation of growing size shifts them), the 16-RHS product of each timed per kernel: does a kernel's time follow where its streams lie?"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import htool_amd as hm
    n, mu = 1000000, 16
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    x = hm.create_geometry("ellipse", n)
    ctb = hm.ClusterTreeBuilder()
    ctb.set_maximal_leaf_size(100)
    T = ctb.create_cluster_tree(n, 3, x, 2, 2)
    hm.lib().hmx_device_init(0)
    free_b, _ = torch.cuda.mem_get_info(0)
    hm.lib().hmx_device_reserve(0, int(0.7 * free_b))
    gen = hm.InvDistGenerator(3, x, x, 1e-5, 1.0, 1.0, 0.0, False)
    rng = np.random.default_rng(1)
    X = torch.from_numpy(rng.random((n, mu))).to(dev)
    Y = torch.zeros((n, mu), dtype=torch.float64, device=dev)
    x1 = torch.from_numpy(rng.random(n)).to(dev)
    y1 = torch.zeros(n, dtype=torch.float64, device=dev)
    keep = []
    for trial in range(int(os.environ.get("HMX_PLACEMENT_BUILDS", "10"))):
        tb = hm.HMatrixTreeBuilder(1e-4, 10.0, "N", "N")
        tb.set_low_rank_generator("partialACA")
        d = bench.minimal_depth(n)
        tb.set_minimal_target_depth(d)
        tb.set_minimal_source_depth(d)
        tb.set_option("build_timing", 1)
        H = tb.build(gen, T, T, -1, -1, device=0, dtype=np.float64)
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
        print("[placement] build %d: %s" % (trial, json.dumps(out)), flush=True)
        if trial % 2 == 0:
            keep.append(H)  # its streams stay where they are: the next build lies behind them
        else:
            del H
        keep.append(torch.empty((37 + 101 * trial) << 20, dtype=torch.uint8, device=dev))
        hm.lib().hmx_device_reserve(0, (3 + 2 * trial) << 21)  # a small slab of its own: odd multiples of 2 MiB shift what follows


if __name__ == "__main__":
    main()
