"""Distribution of R-stream chunk widths at N=1e6 (weighted by coefficients): how full are the reduce kernel's waves?"""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import htool_amd as hm
n = 1000000
world = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # >0: the block rows of rank 0 of a `world`-rank run
x = hm.create_geometry("ellipse", n)
b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(100)
T = b.create_cluster_tree(n, 3, x, 2, world if world else 2)
tb = hm.HMatrixTreeBuilder(1e-4, 10.0, "N", "N"); tb.set_low_rank_generator("partialACA")
tb.set_minimal_target_depth(5); tb.set_minimal_source_depth(5)
H = tb.build(hm.InvDistGenerator(3, x, x, 1e-5, 1.0), T, T, 0 if world else -1, 0 if world else -1)
lt = H.leaf_table()
lr = lt[lt[:, 4] > 0]
key = lr[:, 2].astype(np.int64) * (1 << 32) + lr[:, 3]
uk, inv = np.unique(key, return_inverse=True)
C = np.bincount(inv, weights=lr[:, 4]).astype(np.int64)
size = (uk & 0xFFFFFFFF).astype(np.int64)
hist = {}
tot = 0
for c, s in zip(C, size):
    nch = (c + 127) // 128
    cw = (((c + nch - 1) // nch) + 1) & ~1
    for k in range(nch):
        w = min(cw, c - k * cw)
        hist[w // 16] = hist.get(w // 16, 0) + w * s
        tot += w * s
print("distinct source clusters", len(uk), "reduce coeffs", tot)
for k in sorted(hist):
    print("width %3d-%3d: %5.1f %%" % (16 * k, 16 * k + 15, 100.0 * hist[k] / tot))
print("C quantiles", np.percentile(C, [5, 25, 50, 75, 95]), "size quantiles", np.percentile(size, [5, 25, 50, 75, 95]))
