"""DistributedOperator logic (partitions, collectives, numbering) on CPU with the gloo backend,
world_size 2 and 4.  The local operator is the CPU oracle's rank-restricted H-matrix (the HIP engine
needs a GPU); the distributed products must equal the single-process product of the whole operator,
for trans N/T, alpha/beta, global-to-global (internal and user numbering) and local-to-local
(tests/functional_tests/distributed_operator/test_distributed_operator.hpp:129-170 checks the same)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import htool_amd as hm
from htool_amd import distributed as D
from helpers import rel_err


class OracleLocalOperator:
    """Test stand-in for RestrictedGlobalToLocalHMatrix backed by the CPU oracle."""

    def __init__(self, H):
        self.H = H

    def add_vector_product(self, trans, alpha, x, beta, y):
        out = self.H.matvec(x.numpy(), trans, alpha, beta, y.numpy())
        y.copy_(torch.from_numpy(out))

    def add_matrix_product_row_major(self, trans, alpha, X, beta, Y, mu):
        out = self.H.matmat_row_major(X.numpy(), trans, alpha, beta, Y.numpy())
        Y.copy_(torch.from_numpy(out))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sym, uplo, q, collective="allgather"):
    from oracle import oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1500
        x = hm.create_geometry("ball", n)
        b = hm.ClusterTreeBuilder()
        b.set_maximal_leaf_size(40)
        T = b.create_cluster_tree(n, 3, x, 2, world)  # product host structure
        To = O.ClusterTree(x, 40, 2, world)
        comp = "sympartialACA" if sym == "S" else "partialACA"
        Hloc = O.HMatrix(To, To, eps=1e-6, eta=10.0, sym=sym, uplo=uplo, compressor=comp, rank=rank)
        Hfull = O.HMatrix(To, To, eps=1e-6, eta=10.0, sym=sym, uplo=uplo, compressor=comp)
        tp = D.PartitionFromCluster(T)
        A = D.DistributedOperator(tp, tp, output_collective=collective)
        A.add_global_to_local_operator(OracleLocalOperator(Hloc))
        perm = T.get_permutation()
        xin, y0 = O.hashed_vector(n, 7), O.hashed_vector(n, 8)
        errs = []
        for trans in ("N", "T"):
            # internal numbering
            y = torch.from_numpy(y0.copy())
            D.internal_add_distributed_operator_vector_product_global_to_global(trans, 3.0, A, torch.from_numpy(xin), 2.0, y)
            errs.append(rel_err(y.numpy(), Hfull.matvec(xin, trans, 3.0, 2.0, y0)))
            # user numbering
            y = torch.from_numpy(y0.copy())
            D.add_distributed_operator_vector_product_global_to_global(trans, 3.0, A, torch.from_numpy(xin), 2.0, y)
            yc = Hfull.matvec(xin[perm], trans, 3.0, 2.0, y0[perm])
            yu = np.empty(n)
            yu[perm] = yc
            errs.append(rel_err(y.numpy(), yu))
            # local to local
            off, sz = tp.get_offset_of_partition(rank), tp.get_size_of_partition(rank)
            yl = torch.from_numpy(y0[off:off + sz].copy())
            D.internal_add_distributed_operator_vector_product_local_to_local(trans, 3.0, A, torch.from_numpy(xin[off:off + sz].copy()), 2.0, yl)
            errs.append(rel_err(yl.numpy(), Hfull.matvec(xin, trans, 3.0, 2.0, y0)[off:off + sz]))
            # beta = 0 path
            y = torch.from_numpy(y0.copy())
            D.internal_add_distributed_operator_vector_product_global_to_global(trans, 1.0, A, torch.from_numpy(xin), 0.0, y)
            errs.append(rel_err(y.numpy(), Hfull.matvec(xin, trans, 1.0, 0.0)))
        if sym == "N":  # multi-RHS row-major, mu = 3
            X, Y0 = O.hashed_vector(3 * n, 9).reshape(n, 3), O.hashed_vector(3 * n, 10).reshape(n, 3)
            for trans in ("N", "T"):
                Y = torch.from_numpy(Y0.copy())
                D.internal_add_distributed_operator_matrix_product_row_major_global_to_global(trans, 3.0, A, torch.from_numpy(X.copy()), 2.0, Y, 3)
                errs.append(rel_err(Y.numpy(), Hfull.matmat_row_major(X, trans, 3.0, 2.0, Y0)))
        # block-diagonal operator (DefaultLocalApproximationBuilder): local-to-local operators only
        Hdiag = O.HMatrix(To, To, eps=1e-6, eta=10.0, sym=sym, uplo=uplo, compressor=comp, root_partition=rank)
        B = D.DistributedOperator(tp, tp)
        B.add_local_to_local_operator(OracleLocalOperator(Hdiag))
        off, sz = tp.get_offset_of_partition(rank), tp.get_size_of_partition(rank)
        for trans in ("N", "T"):
            y = torch.from_numpy(y0.copy())
            D.internal_add_distributed_operator_vector_product_global_to_global(trans, 3.0, B, torch.from_numpy(xin), 2.0, y)
            # every rank's diagonal block acts on its own slice; the off-diagonal part of y is beta * y0
            ref_loc = Hdiag.matvec(xin[off:off + sz], trans, 3.0, 2.0, y0[off:off + sz])
            errs.append(rel_err(y.numpy()[off:off + sz], ref_loc))
            yl = torch.from_numpy(y0[off:off + sz].copy())
            D.internal_add_distributed_operator_vector_product_local_to_local(trans, 3.0, B, torch.from_numpy(xin[off:off + sz].copy()), 2.0, yl)
            errs.append(rel_err(yl.numpy(), ref_loc))
        q.put((rank, max(errs)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,sym,uplo,collective", [(2, "N", "N", "allgather"), (4, "N", "N", "allgather"), (2, "S", "L", "allgather"),
                                                       (2, "S", "U", "allgather"), (3, "N", "N", "allreduce")])
def test_distributed_products_match_single_process(world, sym, uplo, collective):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sym, uplo, q, collective)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    # same compressed blocks on both sides except across partition boundaries of the block tree, where the
    # rank-restricted tree may split differently: tolerance = compression accuracy (1e-6), as the reference
    assert max(res.values()) < 1e-5, res


def _zworker(rank, world, port, sym, uplo, q):
    """Complex coefficients: Hermitian / complex symmetric row slabs, trans N / T / C, all-gather and all-reduce of complex
    vectors (through their real views)."""
    from oracle import oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1200
        x = hm.create_geometry("ball", n)
        b = hm.ClusterTreeBuilder()
        b.set_maximal_leaf_size(40)
        T = b.create_cluster_tree(n, 3, x, 2, world)
        To = O.ClusterTree(x, 40, 2, world)
        comp = "sympartialACA" if sym != "N" else "partialACA"
        kw = dict(cre=0.8, cim=0.6, eps=1e-6, eta=10.0, sym=sym, uplo=uplo, compressor=comp)
        Hloc, Hfull = O.ZHMatrix(To, To, rank=rank, **kw), O.ZHMatrix(To, To, **kw)
        tp = D.PartitionFromCluster(T)
        A = D.DistributedOperator(tp, tp)
        A.add_global_to_local_operator(OracleLocalOperator(Hloc))
        xin, y0 = O.hashed_zvector(n, 7), O.hashed_zvector(n, 8)
        alpha, beta = 3.0 + 0.5j, 2.0 - 0.25j
        perm = T.get_permutation()
        errs = []
        for trans in ("N",) + (("T",) if sym != "H" else ()) + (("C",) if sym != "S" else ()):
            y = torch.from_numpy(y0.copy())
            D.internal_add_distributed_operator_vector_product_global_to_global(trans, alpha, A, torch.from_numpy(xin), beta, y)
            errs.append(rel_err(y.numpy(), Hfull.matvec(xin, trans, alpha, beta, y0)))
            y = torch.from_numpy(y0.copy())
            D.add_distributed_operator_vector_product_global_to_global(trans, alpha, A, torch.from_numpy(xin), beta, y)
            yu = np.empty(n, dtype=np.complex128)
            yu[perm] = Hfull.matvec(xin[perm], trans, alpha, beta, y0[perm])
            errs.append(rel_err(y.numpy(), yu))
            off, sz = tp.get_offset_of_partition(rank), tp.get_size_of_partition(rank)
            yl = torch.from_numpy(y0[off:off + sz].copy())
            D.internal_add_distributed_operator_vector_product_local_to_local(trans, alpha, A, torch.from_numpy(xin[off:off + sz].copy()), beta, yl)
            errs.append(rel_err(yl.numpy(), Hfull.matvec(xin, trans, alpha, beta, y0)[off:off + sz]))
        q.put((rank, max(errs)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,sym,uplo", [(2, "N", "N"), (2, "H", "L"), (2, "S", "U")])
def test_distributed_complex_products(world, sym, uplo):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zworker, args=(r, world, port, sym, uplo, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert max(res.values()) < 1e-5, res


def test_partition_numbering_round_trip():
    x = hm.create_geometry("disk", 500)
    b = hm.ClusterTreeBuilder()
    T = b.create_cluster_tree(500, 3, x, 2, 4)
    p = D.PartitionFromCluster(T)
    assert p.number_of_partitions() == 4 and p.get_global_size() == 500
    assert sum(p.get_size_of_partition(k) for k in range(4)) == 500
    v = torch.arange(500, dtype=torch.float64)
    w = p.global_to_partition_numbering(v)
    assert torch.equal(w, torch.from_numpy(T.get_permutation().astype(np.float64)))
    assert torch.equal(p.partition_to_global_numbering(w), v)


class _ReferenceLocalHMatrix:
    """Rank-local H-matrix described by the reference's own leaf table for that rank (fixture *_rank<k>): what
    get_distributed_hmatrix_information needs from an HMatrix, without a GPU."""

    def __init__(self, fixture):
        from helpers import load
        g = load(fixture)
        self._leaves, self._root = g["leaves"], g["rootinfo"]

    def nb_rows(self):
        return int(self._root[1])

    def nb_cols(self):
        return int(self._root[3])

    def leaf_table(self):
        return self._leaves

    def stats(self):
        return dict(n_false_positive=int(self._root[4]), t_compress_s=0.01 * (1 + int(self._root[0] > 0)), t_assemble_s=0.0, t_pack_s=0.0)


def _info_worker(rank, world, port, stem, q):
    import io
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = io.StringIO()
        D.print_distributed_hmatrix_information(_ReferenceLocalHMatrix("%s_rank%d" % (stem, rank)), out)
        q.put((rank, out.getvalue()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("stem,fixture,world", [("ellipse_n4000_p4", "distinfo_ellipse_n4000_p4", 4), ("ball_n2000_p2_symL", "distinfo_ball_n2000_p2_symL", 2)])
def test_distributed_hmatrix_information_matches_reference_text(stem, fixture, world):
    """print_distributed_hmatrix_information over gloo, fed with the reference's per-rank leaf tables, against the text the
    reference printed under MPI for the same operator (tests/golden/distinfo_*: oracle/_ref/dist_info); its wall-clock and
    OpenMP lines are machine facts and are left out of the comparison."""
    from helpers import load
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_info_worker, args=(r, world, port, stem, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    skip = ("Block_tree_walltime", "Blocks_computation_walltime", "Number_of_threads_per_tasks", "Number_of_procs")

    def lines(t):
        return [ln for ln in t.splitlines() if not ln.startswith(skip)]

    ref = load(fixture)["information"].tobytes().decode()
    assert lines(got[0]) == lines(ref)
    assert all(got[r] == "" for r in range(1, world))  # rank 0 prints, as in the reference
    timing = [ln for ln in got[0].splitlines() if ln.startswith("Blocks_computation_walltime")]
    assert len(timing) == 3 and all(ln.endswith(" second(s)") for ln in timing)
