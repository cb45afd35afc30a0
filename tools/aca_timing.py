import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, htool_amd as hm
n = 1000000
x = hm.create_geometry("ellipse", n)
b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(100)
T = b.create_cluster_tree(n, 3, x, 2, 2)
tb = hm.HMatrixTreeBuilder(1e-4, 10.0, "N", "N", int(os.environ.get("ACA_REQRANK", "-1"))); tb.set_low_rank_generator("partialACA")
tb.set_minimal_target_depth(6); tb.set_minimal_source_depth(6)
for rep in range(2):
    H = tb.build(hm.InvDistGenerator(3, x, x, 1e-5, 1.0), T, T)
    st = H.stats()
    print("team", os.environ.get("HMX_ACA_TEAM"), os.environ.get("HMX_ACA_TEAM_MIN"), os.environ.get("HMX_ACA_TEAM_Q"), "ACA %.1f ms" % (1e3 * st["t_compress_s"]), flush=True)
    del H
