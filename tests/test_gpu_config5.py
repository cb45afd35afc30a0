"""BASELINE config 5's exact flags on ONE GPU: N=4e6 fp32 coefficients, 'S','L' storage, sympartialACA eps=1e-6, 16 right-hand sides
(row-major) -- the per-rank operator of the 8-GPU configuration is a row restriction of this one.  The product is checked against the
CPU oracle's row-major product (add_hmatrix_matrix_product_row_major.hpp:58-178 restated) on the block rows of the LAST partition-
aligned cluster of the rows, multiplying the blocks the engine itself compressed: every stored leaf of those rows is downloaded, the
mirrored contributions come from the leaves inside the slab (rows further up only contribute to rows above the slab).  Also the
single-vector fused symmetric product on the same operator against column 0.  HMX_TEST_C5_N shrinks it."""
import os

import numpy as np
import pytest

import bench
import htool_amd as hm
from helpers import rel_err

pytestmark = pytest.mark.gpu


def test_config5_flags_one_gpu_sixteen_rhs_against_oracle_on_a_row_slab():
    from oracle import oracle as O
    # the 16-RHS product on the STORED TRIANGLE (default when HBM has no room for an expanded copy; forced here: one GPU has)
    n, mu = int(os.environ.get("HMX_TEST_C5_N", 4000000)), 16
    x = hm.create_geometry("ellipse", n)
    b = hm.ClusterTreeBuilder()
    b.set_maximal_leaf_size(100)
    T = b.create_cluster_tree(n, 3, x, 2, 2)
    tb = hm.HMatrixTreeBuilder(1e-6, 10.0, "S", "L")
    tb.set_low_rank_generator("sympartialACA")
    d = bench.minimal_depth(n)
    tb.set_minimal_target_depth(d)
    tb.set_minimal_source_depth(d)
    H = tb.build(hm.InvDistGenerator(3, x, x, 1e-5, 1.0), T, T, dtype=np.float32)
    H.set_option("sym_multi_rhs", 1)
    st = H.stats()
    assert st["n_false_positive"] == 0 and st["rank_max"] < 64
    rng = np.random.default_rng(5)
    X = rng.random((n, mu)).astype(np.float32)
    Y = np.zeros((n, mu), dtype=np.float32)
    hm.internal_add_hmatrix_matrix_product_row_major("N", 1.0, H, X, 0.0, Y, mu)
    # the 16-RHS product ran on the STORED TRIANGLE: no expanded copy of the operator was allocated (round 3: 42 GB + 83 GB on this GPU)
    after = H.stats()
    assert after["expanded_bytes"] == 0 and after["stream_bytes"] == st["stream_bytes"]
    assert n < 4000000 or 38e9 < after["stream_bytes"] < 46e9
    # the slab: the smallest cluster ending at n that holds whole leaves and at least n / 256 rows
    tab = np.asarray(H.leaf_table())
    nodes = T.nodes_int()
    need = max(int(tab[:, 1].max()), n // 256)
    cand = nodes[(nodes[:, 1] + nodes[:, 2] == n) & (nodes[:, 2] >= need)]
    cut = int(cand[:, 2].min())
    lo = n - cut
    sel = np.nonzero(tab[:, 0] >= lo)[0]
    assert np.all(tab[sel, 0] + tab[sel, 1] <= n) and len(sel) > 100
    data, offs, pos = [], [], 0
    for leaf in sel:
        blk = H.get_block(int(leaf))
        if tab[leaf, 4] >= 0:
            u, v = np.asfortranarray(blk[0], dtype=np.float64).ravel("F"), np.asfortranarray(blk[1], dtype=np.float64).ravel("F")
            offs.append((pos, pos + u.size))
            data += [u, v]
            pos += u.size + v.size
        else:
            dd = np.asfortranarray(blk, dtype=np.float64).ravel("F")
            offs.append((pos, 0))
            data.append(dd)
            pos += dd.size
    desc = tab[sel].copy()
    desc[desc[:, 2] < lo, 5] = 0  # leaves whose columns lie left of the slab mirror into rows ABOVE the slab: not part of the check
    Ho = O.HMatrix.from_blocks(desc, np.array(offs), np.concatenate(data), [lo, cut, 0, n], "S", "L")
    ref = Ho.matmat_row_major(X.astype(np.float64))
    err = rel_err(Y[lo:].astype(np.float64), ref)
    print("config 5 flags, N=%d: %d + %d leaves, rank %d/%.2f/%d, %.1f GB of streams; slab rows [%d, %d): %d leaves (%.2f GB), mu=%d error vs oracle %.2e"
          % (n, st["n_dense"], st["n_lowrank"], st["rank_min"], st["rank_mean"], st["rank_max"], st["stream_bytes"] / 1e9, lo, n, len(sel), pos * 8 / 1e9, mu, err))
    assert err < 2e-5  # fp32 arithmetic on the same blocks (the fp32 parity bar)
    # the fused single-vector product on compact storage = column 0 of the 16-RHS product (both on the stored triangle)
    y = np.zeros(n, dtype=np.float32)
    hm.internal_add_hmatrix_vector_product("N", 1.0, H, np.ascontiguousarray(X[:, 0]), 0.0, y)
    assert rel_err(y.astype(np.float64), Y[:, 0].astype(np.float64)) < 2e-5
