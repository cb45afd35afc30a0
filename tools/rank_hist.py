"""Rank distribution of one bench configuration's low-rank leaves: where the ACA kernel's work sits (sum over blocks of rank^2 (m+n)).
usage: python tools/rank_hist.py [--dtype z64 --sym H ...bench flags for n/eps/eta/leaf]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import htool_amd as hm
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--dtype", default="z64")
ap.add_argument("--sym", default="H")
ap.add_argument("--eps", type=float, default=1e-4)
a = ap.parse_args()
n = a.n
x = hm.create_geometry("ellipse", n)
b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(100)
T = b.create_cluster_tree(n, 3, x, 2, 2)
sym = a.sym
tb = hm.HMatrixTreeBuilder(a.eps, 10.0, sym, "L" if sym != "N" else "N")
d = bench.minimal_depth(n)
tb.set_minimal_target_depth(d); tb.set_minimal_source_depth(d)
np_dt = {"f64": np.float64, "f32": np.float32, "z64": np.complex128, "c32": np.complex64}[a.dtype]
cplx = a.dtype in ("z64", "c32")
gen = hm.InvDistGenerator(3, x, x, 1e-5, 1.0, 1.0, 1.0 if cplx else 0.0, sym == "H")
t = time.time(); H = tb.build(gen, T, T, -1, -1, dtype=np_dt); print("build %.2f s" % (time.time() - t))
tab = np.asarray(H.leaf_table())
lr = tab[tab[:, 4] > 0]
m, nn, r = lr[:, 1].astype(np.int64), lr[:, 3].astype(np.int64), lr[:, 4].astype(np.int64)
w = r * r * (m + nn)
print("low-rank leaves %d, sum rank^2 (m+n) = %.3e scalars" % (len(lr), w.sum()))
for lo, hi in [(1, 16), (16, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 4096)]:
    s = (r >= lo) & (r < hi)
    if s.any():
        print("rank [%4d,%4d): %7d blocks, m+n mean %8.0f max %8d, share of work %.3f, heaviest block %.3e" % (lo, hi, s.sum(), (m + nn)[s].mean(), (m + nn)[s].max(), w[s].sum() / w.sum(), w[s].max()))
o = np.argsort(-w)[:12]
for i in o:
    print("  block %dx%d rank %d  work %.3e" % (m[i], nn[i], r[i], w[i]))
