"""Error of the compressed operator against exact kernel rows for the sign-discontinuous Hermitian generator (ACA's stopping estimate is a
heuristic; on a discontinuous kernel it stops above epsilon -- in the reference as well: the N=1e5 operator is rank-identical to htool's)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import htool_amd as hm
for n, depth in ((100000, 3), (1000000, 5)):
    x = hm.create_geometry("ellipse", n)
    b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(100)
    T = b.create_cluster_tree(n, 3, x, 2, 2)
    for sym, gen_h in (("H", True), ("N", True), ("N", False)):
        tb = hm.HMatrixTreeBuilder(1e-4, 10.0, sym, "L" if sym == "H" else "N")
        tb.set_low_rank_generator("sympartialACA" if sym == "H" else "partialACA")
        tb.set_minimal_target_depth(depth); tb.set_minimal_source_depth(depth)
        H = tb.build(hm.InvDistGenerator(3, x, x, 1e-5, 1.0, 1.0, 1.0, gen_h), T, T, dtype=np.complex128)
        rng = np.random.default_rng(7)
        u = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        y = np.zeros(n, dtype=np.complex128)
        hm.add_hmatrix_vector_product("N", 1.0, H, u, 0.0, y)
        rows = rng.choice(n, 48, replace=False)
        exact = np.empty(len(rows), dtype=np.complex128)
        for k, i in enumerate(rows):
            d = np.sqrt(((x[i][None, :] - x) ** 2).sum(-1))
            sg = np.sign(x[i, 0] - x[:, 0]) if gen_h else 1.0
            exact[k] = ((1.0 + 1j * sg) / (1e-5 + d)) @ u
        print("n=%d storage %s generator %s: max rank %d, error against exact rows %.2e" % (n, sym, "sign-discontinuous" if gen_h else "smooth", int(np.asarray(H.leaf_table())[:, 4].max()),
              np.linalg.norm(y[rows] - exact) / np.linalg.norm(exact)), flush=True)
        del H
