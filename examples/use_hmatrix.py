#!/usr/bin/env python3
"""The reference's examples/use_hmatrix.cpp (lines 77-111) on the HIP engine, same parameters: N = 10 000 points on a
planar 4:1 ellipse, leaf size 500, Partitioning_N<ComputeLargestExtent, RegularSplitting>, eps = 0.01, eta = 200,
symmetric 'S','L' storage, default compressor (sympartialACA), kernel 1/(1e-5 + |x-y|), x = 1.
The reference prints "relative error on matrix vector product : 2.67e-04" for this setup
and reports 107 dense + 82 low-rank leaves."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import htool_amd as hm  # noqa: E402


def run(device=0, output_folder=None):
    n = 10000
    coordinates = hm.create_geometry("ellipse", n)
    builder = hm.ClusterTreeBuilder()
    builder.set_maximal_leaf_size(500)
    builder.set_partitioning_strategy("largest_extent", "regular", partitioning_n=True)
    cluster = builder.create_cluster_tree(n, 3, coordinates, 2, 2)  # HMatrixBuilder: 2 children, 2 partitions (hmatrix/utility.hpp:23)
    A = hm.InvDistGenerator(3, coordinates, coordinates, 1e-5, 1.0)
    hmatrix = hm.HMatrixTreeBuilder(0.01, 200.0, "S", "L").build(A, cluster, cluster, device=device)
    x, y = np.ones(n), np.zeros(n)
    hm.add_hmatrix_vector_product("N", 1.0, hmatrix, x, 0.0, y)
    d = np.sqrt(((coordinates[:, None, :] - coordinates[None, :, :]) ** 2).sum(-1))
    ref = (1.0 / (1e-5 + d)) @ x
    st = hmatrix.stats()
    err = np.linalg.norm(ref - y) / np.linalg.norm(ref)
    if output_folder is not None:  # what use_hmatrix.cpp:101-103 writes and prints
        hm.save_leaves_with_rank(hmatrix, os.path.join(output_folder, "hmatrix"))
        hm.print_tree_parameters(hmatrix)
        hm.print_hmatrix_information(hmatrix)
    return err, st


if __name__ == "__main__":
    if len(sys.argv) > 2:
        sys.exit("Usage: %s output_folder" % sys.argv[0])
    err, st = run(output_folder=sys.argv[1] if len(sys.argv) == 2 else "./")
    print("dense leaves %d, low-rank leaves %d, rank %d/%.2f/%d" % (st["n_dense"], st["n_lowrank"], st["rank_min"], st["rank_mean"], st["rank_max"]))
    print("relative error on matrix vector product : %.3e" % err)
