#!/usr/bin/env python3
"""The reference's examples/use_distributed_operator.cpp (lines 74-117) on the HIP engine, same parameters: N = 10 000
(planar ellipse), 8 children per cluster, one partition per rank, leaf size 100, Partitioning_N, eps = 1e-3, eta = 100,
'S','U', kernel 1/(1 + |x-y|), x = 1, global-to-global product in user numbering.
Run under torch.distributed (one process per GPU):
    python -m torch.distributed.run --nproc-per-node P --master-addr 127.0.0.1 examples/use_distributed_operator.py
or in a single process with --emulate P (ranks built one after the other on one GPU).
The reference prints 9.8e-5 / 9.3e-5 / 6.6e-5 for P = 1 / 2 / 4."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import htool_amd as hm  # noqa: E402
from htool_amd import distributed as D


def setup(P):
    n = 10000
    coordinates = hm.create_geometry("ellipse", n)
    b = hm.ClusterTreeBuilder()
    b.set_maximal_leaf_size(100)
    b.set_partitioning_strategy("largest_extent", "regular", partitioning_n=True)
    cluster = b.create_cluster_tree(n, 3, coordinates, 8, P)
    A = hm.InvDistGenerator(3, coordinates, coordinates, 1.0, 1.0)
    tb = hm.HMatrixTreeBuilder(1e-3, 100.0, "S", "U")
    d = np.sqrt(((coordinates[:, None, :] - coordinates[None, :, :]) ** 2).sum(-1))
    ref = (1.0 / (1.0 + d)) @ np.ones(n)
    return n, cluster, A, tb, ref


def run_emulated(P, device=0):
    """All P ranks in one process: rank k's local operator produces its slice of y (no collective needed)."""
    n, cluster, A, tb, ref = setup(P)
    perm = cluster.get_permutation()
    x_part = np.ones(n)[perm]  # global_to_partition_numbering
    y_part = np.zeros(n)
    for k in range(P):
        H = tb.build(A, cluster, cluster, k, k, device=device)
        yk = np.zeros(H.nb_rows())
        hm.internal_add_hmatrix_vector_product("N", 1.0, H, x_part, 0.0, yk)
        y_part[H.target_offset:H.target_offset + H.nb_rows()] = yk
    y = np.empty(n)
    y[perm] = y_part  # partition_to_global_numbering
    return np.linalg.norm(ref - y) / np.linalg.norm(ref)


def run_distributed():
    # both must be in the environment before HIP / HSA initialise (the first cuda call below)
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # single node: RCCL bootstrap over loopback
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (the only mode the platform's host driver supports)
    import torch
    import torch.distributed as dist
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if "MASTER_ADDR" not in os.environ:  # plain `python examples/use_distributed_operator.py`: one rank
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n, cluster, A, tb, ref = setup(dist.get_world_size())
    builder = D.DefaultApproximationBuilder(A, cluster, cluster, tb, device=local_rank)
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    D.add_distributed_operator_vector_product_global_to_global("N", 1.0, builder.distributed_operator, x, 0.0, y)
    err = np.linalg.norm(ref - y.cpu().numpy()) / np.linalg.norm(ref)
    if dist.get_rank() == 0:
        print("relative error on global to global matrix vector product : %.3e" % err)
    # what use_distributed_operator.cpp:118-126 prints and writes
    local_hmatrix = builder.hmatrix
    if dist.get_rank() == 0:
        hm.print_tree_parameters(local_hmatrix)
        hm.print_hmatrix_information(local_hmatrix)
    D.print_distributed_hmatrix_information(local_hmatrix)
    out = [a for a in sys.argv[1:] if not a.startswith("--")]
    hm.save_leaves_with_rank(local_hmatrix, os.path.join(out[0] if out else "./", "local_hmatrix_%d" % dist.get_rank()))
    dist.destroy_process_group()


if __name__ == "__main__":
    if "--emulate" in sys.argv:
        P = int(sys.argv[sys.argv.index("--emulate") + 1])
        print("relative error on global to global matrix vector product (P=%d, emulated) : %.3e" % (P, run_emulated(P)))
    else:
        run_distributed()
