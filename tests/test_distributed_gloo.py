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


# ---- every product family of the DistributedOperator against what htool itself computed under MPI -----------------------------------
# tests/golden/distprod_*: oracle/_ref/dist_products (the REAL reference, mpiexec -n world) ran vector / row-major / column-major x
# global-to-global / local-to-local x user / partition numbering + the sub product on closed-form inputs and stored every rank's output.
DISTPROD_CASES = sorted(k for k, v in __import__("helpers").MANIFEST.items() if v["mode"] == "distprod")


class OracleOperatorWithSubProduct:
    """OracleLocalOperator + add_sub_matrix_product_to_local, for real and complex oracle H-matrices (numpy <-> torch, CPU)."""

    def __init__(self, H):
        self.H = H
        self.source_offset, self.source_size = int(H.rootinfo[2]), int(H.rootinfo[3])

    def add_vector_product(self, trans, alpha, x, beta, y):
        y.copy_(torch.from_numpy(self.H.matvec(x.numpy(), trans, alpha, beta, y.numpy())))

    def add_matrix_product_row_major(self, trans, alpha, X, beta, Y, mu):
        Y.copy_(torch.from_numpy(self.H.matmat_row_major(X.contiguous().numpy(), trans, alpha, beta, Y.contiguous().numpy())))

    def add_sub_matrix_product_to_local(self, X, Y, mu, offset, size):
        D.add_sub_matrix_product_to_local(self, self.source_offset, self.source_size, X, Y, mu, offset, size)


def distprod_inputs(n, mu, cplx):
    """The closed-form inputs of oracle/ref/dist_products.cpp: vectors (salts 17, 18) and column-major n x mu matrices (21, 22)."""
    from oracle import oracle as O
    h = O.hashed_zvector if cplx else O.hashed_vector
    return h(n, 17), h(n, 18), h(n * mu, 21).reshape(mu, n).T, h(n * mu, 22).reshape(mu, n).T  # (n, mu) views of column-major data


def run_distprod_families(A, rank, world, g, p, mk, native=None):
    """Runs every family on operator A (and, when given, the C-level NativeDistributedOperator `native`) and returns
    {name: relative error against the reference's MPI output}.  `mk` turns a numpy array into the tensor type under test."""
    n, mu, cplx = p["n"], p["mu"], p.get("prec") == "z64"
    ab = g["alpha_beta"]
    alpha, beta = (complex(ab[0], ab[1]), complex(ab[2], ab[3])) if cplx else (float(ab[0]), float(ab[2]))
    xin, y0, X, Y0 = distprod_inputs(n, mu, cplx)
    off, sz = int(g["partition"][rank, 0]), int(g["partition"][rank, 1])
    local_numbering = bool(g["r%d_world_rank_n_mu_localnumbering" % rank][4])
    sym = p.get("sym", "N")
    errs = {}

    def cm(M):  # a column-major device / host matrix of shape (rows, mu); always a copy (torch.from_numpy shares memory)
        return mk(np.array(M.T, order="C", copy=True)).T

    def ref(name, r=0):
        return g["r%d_%s" % (r, name)]

    transes = "NT" if not cplx else ("NC" if sym == "H" else ("NT" if sym == "S" else "NTC"))
    for t in transes:
        # vectors
        y = mk(y0.copy())
        D.add_distributed_operator_vector_product_global_to_global(t, alpha, A, mk(xin), beta, y)
        errs["g2g_user_" + t] = rel_err(y.cpu().numpy(), ref("g2g_user_" + t))
        y = mk(y0.copy())
        D.internal_add_distributed_operator_vector_product_global_to_global(t, alpha, A, mk(xin), beta, y)
        errs["g2g_internal_" + t] = rel_err(y.cpu().numpy(), ref("g2g_internal_" + t))
        yl = mk(y0[off:off + sz].copy())
        D.internal_add_distributed_operator_vector_product_local_to_local(t, alpha, A, mk(xin[off:off + sz].copy()), beta, yl)
        errs["l2l_internal_" + t] = rel_err(yl.cpu().numpy(), ref("l2l_internal_" + t, rank))
        if local_numbering:
            yl = mk(y0[off:off + sz].copy())
            D.add_distributed_operator_vector_product_local_to_local(t, alpha, A, mk(xin[off:off + sz].copy()), beta, yl)
            errs["l2l_user_" + t] = rel_err(yl.cpu().numpy(), ref("l2l_user_" + t, rank))
        # row-major multi-RHS
        Y = mk(np.ascontiguousarray(Y0))
        D.internal_add_distributed_operator_matrix_product_row_major_global_to_global(t, alpha, A, mk(np.ascontiguousarray(X)), beta, Y, mu)
        errs["g2g_rm_" + t] = rel_err(Y.cpu().numpy(), ref("g2g_rm_" + t))
        for name, b in (("l2l_rm_", beta), ("l2l_rm_beta0_", 0.0)):
            Yl = mk(np.ascontiguousarray(Y0[off:off + sz]))
            D.internal_add_distributed_operator_matrix_product_row_major_local_to_local(t, alpha, A, mk(np.ascontiguousarray(X[off:off + sz])), b, Yl, mu)
            errs[name + t] = rel_err(Yl.cpu().numpy(), ref(name + t, rank))
        # column-major multi-RHS (fixtures hold the column-major data as (mu, rows))
        Y = cm(Y0)
        D.add_distributed_operator_matrix_product_global_to_global(t, alpha, A, cm(X), beta, Y)
        errs["g2g_cm_user_" + t] = rel_err(Y.T.cpu().numpy(), ref("g2g_cm_user_" + t))
        Y = cm(Y0)
        D.internal_add_distributed_operator_matrix_product_global_to_global(t, alpha, A, cm(X), beta, Y)
        errs["g2g_cm_internal_" + t] = rel_err(Y.T.cpu().numpy(), ref("g2g_cm_internal_" + t))
        Y = cm(Y0)
        D.add_distributed_operator_matrix_product_global_to_global(t, alpha, A, cm(X), 0.0, Y)
        errs["g2g_cm_user_beta0_" + t] = rel_err(Y.T.cpu().numpy(), ref("g2g_cm_user_beta0_" + t))
        Yl = cm(Y0[off:off + sz])
        D.internal_add_distributed_operator_matrix_product_local_to_local(t, alpha, A, cm(X[off:off + sz]), beta, Yl)
        errs["l2l_cm_internal_" + t] = rel_err(Yl.T.cpu().numpy(), ref("l2l_cm_internal_" + t, rank))
        if local_numbering:
            Yl = cm(Y0[off:off + sz])
            D.add_distributed_operator_matrix_product_local_to_local(t, alpha, A, cm(X[off:off + sz]), beta, Yl)
            errs["l2l_cm_user_" + t] = rel_err(Yl.T.cpu().numpy(), ref("l2l_cm_user_" + t, rank))
        if native is not None:  # the same products as ONE C call each (hmx_dist_*)
            Yl = mk(np.ascontiguousarray(Y0[off:off + sz]))
            native.matmat_row_major_local_to_local(t, alpha, mk(np.ascontiguousarray(X[off:off + sz])), beta, Yl, mu)
            errs["native_l2l_rm_" + t] = rel_err(Yl.cpu().numpy(), ref("l2l_rm_" + t, rank))
            Yl = mk(np.ascontiguousarray(Y0[off:off + sz]))
            native.matmat_row_major_local_to_local(t, alpha, mk(np.ascontiguousarray(X[off:off + sz])), 0.0, Yl, mu)
            errs["native_l2l_rm_beta0_" + t] = rel_err(Yl.cpu().numpy(), ref("l2l_rm_beta0_" + t, rank))
            for user, tag in ((True, "user"), (False, "internal")):
                Y = cm(Y0)
                native.matmat_global_to_global(t, alpha, cm(X), beta, Y, user_numbering=user)
                errs["native_g2g_cm_%s_%s" % (tag, t)] = rel_err(Y.T.cpu().numpy(), ref("g2g_cm_%s_%s" % (tag, t)))
                if user or local_numbering or True:
                    if user and not local_numbering:
                        continue
                    Yl = cm(Y0[off:off + sz])
                    native.matmat_local_to_local(t, alpha, cm(X[off:off + sz]), beta, Yl, user_numbering=user)
                    errs["native_l2l_cm_%s_%s" % (tag, t)] = rel_err(Yl.T.cpu().numpy(), ref("l2l_cm_%s_%s" % (tag, t), rank))
            Y = cm(Y0)
            native.matmat_global_to_global(t, alpha, cm(X), 0.0, Y, user_numbering=True)
            errs["native_g2g_cm_user_beta0_" + t] = rel_err(Y.T.cpu().numpy(), ref("g2g_cm_user_beta0_" + t))
            y = mk(y0.copy())  # mu = 1: the user-numbering vector products
            native.matmat_global_to_global(t, alpha, mk(xin), beta, y, user_numbering=True)
            errs["native_g2g_user_" + t] = rel_err(y.cpu().numpy(), ref("g2g_user_" + t))
            if local_numbering:
                yl = mk(y0[off:off + sz].copy())
                native.matmat_local_to_local(t, alpha, mk(xin[off:off + sz].copy()), beta, yl, user_numbering=True)
                errs["native_l2l_user_" + t] = rel_err(yl.cpu().numpy(), ref("l2l_user_" + t, rank))
    # HPDDMOperator::GMV's body (wrappers/wrapper_hpddm.hpp:102-142): column-major with leading dimension dof = local size + overlap,
    # alpha = 1, beta = 0, overlap rows of the output zeroed; expected = the reference's beta = 0 row-major product / alpha
    dof = sz + 7
    want = ref("l2l_rm_beta0_N", rank) / alpha
    for m_ in (mu, 1):
        xg = np.zeros((m_, dof), dtype=X.dtype)
        xg[:, :sz] = X[off:off + sz, :m_].T
        for name, fn in (("gmv_mu%d" % m_, lambda x_, y_: D.hpddm_gmv(A, x_, y_, m_, dof)),) + ((("native_gmv_mu%d" % m_, lambda x_, y_: native.gmv(x_, y_, m_, dof)),) if native is not None else ()):
            yg = mk(np.full(m_ * dof, 7.0, dtype=X.dtype))
            fn(mk(xg.ravel().copy()), yg)
            Yg = yg.cpu().numpy().reshape(m_, dof)
            if m_ == mu:
                errs[name] = rel_err(Yg[:, :sz].T, want)
            else:  # mu = 1: the vector kernel; compare with column 0 of a separate single-vector product
                y1 = mk(np.zeros(sz, dtype=X.dtype))
                D.internal_add_distributed_operator_vector_product_local_to_local("N", 1.0, A, mk(np.ascontiguousarray(X[off:off + sz, 0])), 0.0, y1)
                errs[name] = rel_err(Yg[0, :sz], y1.cpu().numpy())
            assert not Yg[:, sz:].any(), name + ": overlap rows of the output must be zero"
    # sub product: the rows of one partition after the other, accumulating (as solvers/geneo/coarse_operator_builder.hpp:99 calls it);
    # then a range that overlaps the partitions partially, mu = 1
    Yl = mk(np.ascontiguousarray(Y0[off:off + sz]))
    for k in range(world):
        o, s_ = int(g["partition"][k, 0]), int(g["partition"][k, 1])
        D.internal_add_distributed_operator_vector_sub_product_global_to_local(A, mk(np.ascontiguousarray(X[o:o + s_])), Yl, mu, o, s_)
    errs["sub_g2l"] = rel_err(Yl.cpu().numpy(), ref("sub_g2l", rank))
    s_off, s_size = (int(v) for v in g["sub_offset_size"])
    yl = mk(y0[off:off + sz].copy()).reshape(-1, 1)
    D.internal_add_distributed_operator_vector_sub_product_global_to_local(A, mk(xin[s_off:s_off + s_size].copy()).reshape(-1, 1), yl, 1, s_off, s_size)
    errs["sub_g2l_partial_mu1"] = rel_err(yl.cpu().numpy().ravel(), ref("sub_g2l_partial_mu1", rank))
    return errs


def distprod_cluster_tree(p, world):
    from oracle import oracle as O
    x = hm.create_geometry(p["geom"], p["n"])
    b = hm.ClusterTreeBuilder()
    b.set_maximal_leaf_size(p["leaf"])
    if p.get("given") == "local":
        part = O.given_partition("local", p["n"], world)
        T = b.create_cluster_tree_from_local_partition(p["n"], 3, x, p.get("children", 2), world, part)
        To = O.ClusterTree(x, p["leaf"], p.get("children", 2), world, given_partition=part, given_local=True)
    else:
        T = b.create_cluster_tree(p["n"], 3, x, p.get("children", 2), world)
        To = O.ClusterTree(x, p["leaf"], p.get("children", 2), world)
    return x, T, To


def _families_worker(rank, world, port, case, q):
    from oracle import oracle as O
    from helpers import load, MANIFEST
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p, g = MANIFEST[case], load(case)
        x, T, To = distprod_cluster_tree(p, world)
        assert np.array_equal(T.get_permutation(), g["perm"]) and np.array_equal(np.asarray(T.get_clusters_on_partition()), g["partition"])
        kw = dict(eps=p["eps"], eta=10.0, sym=p.get("sym", "N"), uplo=p.get("uplo", "N"), compressor=p["compressor"])
        block_diagonal = bool(p.get("local", 0))
        where = dict(root_partition=rank) if block_diagonal else dict(rank=rank)
        Hloc = O.ZHMatrix(To, To, cre=1.0, cim=0.5, **kw, **where) if p.get("prec") == "z64" else O.HMatrix(To, To, **kw, **where)
        assert np.array_equal(Hloc.leaves[:, :5], g["r%d_leaves" % rank])  # this rank's blocks and ranks are htool's
        tp = D.PartitionFromCluster(T)
        A = D.DistributedOperator(tp, tp)
        if block_diagonal:
            A.add_local_to_local_operator(OracleOperatorWithSubProduct(Hloc))
        else:
            A.add_global_to_local_operator(OracleOperatorWithSubProduct(Hloc))
        q.put((rank, run_distprod_families(A, rank, world, g, p, torch.from_numpy)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", DISTPROD_CASES)
def test_every_product_family_matches_the_reference_under_mpi(case):
    """the mu > 1 row-major local-to-local product (HPDDMOperator::GMV, wrappers/wrapper_hpddm.hpp:126) and the
    column-major front ends, with the torch.distributed layer over gloo and the oracle's rank-local operators, against the outputs of
    htool's own MPI run (same world size, same partitions)."""
    from helpers import MANIFEST
    world = MANIFEST[case]["partitions"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_families_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=280) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    worst = {k: max(res[r][k] for r in range(world) if k in res[r]) for r in range(world) for k in res[r]}
    assert len(worst) >= 20
    assert max(worst.values()) < 1e-10, sorted((k, float(v)) for k, v in worst.items() if v >= 1e-10)
