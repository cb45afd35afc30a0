"""The DistributedOperator layer on a real GPU with the RCCL backend (world size 1: the only size one box offers; sizes 2 and 4
are covered on CPU with gloo in test_distributed_gloo.py).  Runs in a subprocess so the process group does not leak into the
rest of the suite.  Checks DefaultApproximationBuilder / DefaultLocalApproximationBuilder, the global-to-global and
local-to-local products in both numberings, the graphed product and complex vectors against the single-operator product."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    import htool_amd as hm
    from htool_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # single node: RCCL bootstrap over loopback
    dist.init_process_group("nccl", device_id=dev)
    rel = lambda a, b: float(torch.linalg.norm(a - b) / torch.linalg.norm(b))
    n = 6000
    x = hm.create_geometry("ellipse", n)
    b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(80)
    T = b.create_cluster_tree(n, 3, x, 2, 1)
    for dtype, tdt, sym, uplo in ((np.float64, torch.float64, "N", "N"), (np.float64, torch.float64, "S", "L"), (np.complex128, torch.complex128, "H", "U")):
        cplx = dtype == np.complex128
        tb = hm.HMatrixTreeBuilder(1e-6, 10.0, sym, uplo)
        tb.set_low_rank_generator("sympartialACA" if sym != "N" else "partialACA")
        gen = hm.InvDistGenerator(3, x, x, 1e-5, 1.0, 1.0, 0.5 if cplx else 0.0, sym == "H")
        B = D.DefaultApproximationBuilder(gen, T, T, tb, dtype=dtype)
        A, H = B.distributed_operator, B.hmatrix
        g = torch.Generator().manual_seed(0)
        xin = torch.randn(n, dtype=tdt, generator=g).to(dev)
        y0 = torch.randn(n, dtype=tdt, generator=g).to(dev)
        alpha, beta = (1.5 - 0.5j, 0.25 + 1j) if cplx else (1.5, 0.25)
        for trans in ("N",) + (("T",) if sym != "H" else ()) + (("C",) if cplx and sym != "S" else ()):
            ref = y0.clone()
            hm.internal_add_hmatrix_vector_product(trans, alpha, H, xin, beta, ref)
            y = y0.clone()
            D.internal_add_distributed_operator_vector_product_global_to_global(trans, alpha, A, xin, beta, y)
            assert rel(y, ref) < 1e-13, ("g2g", sym, trans, rel(y, ref))
            y = y0.clone()
            D.internal_add_distributed_operator_vector_product_local_to_local(trans, alpha, A, xin, beta, y)
            assert rel(y, ref) < 1e-13, ("l2l", sym, trans, rel(y, ref))
            ref_u = y0.clone()
            hm.add_hmatrix_vector_product(trans, alpha, H, xin, beta, ref_u)
            y = y0.clone()
            D.add_distributed_operator_vector_product_global_to_global(trans, alpha, A, xin, beta, y)
            assert rel(y, ref_u) < 1e-13, ("user", sym, trans, rel(y, ref_u))
        # graphed product: local kernels replayed from a HIP graph, collective eager
        yg = torch.zeros(n, dtype=tdt, device=dev)
        gp = D.GraphedGlobalToGlobalProduct(A, xin, yg)
        ref = torch.zeros(n, dtype=tdt, device=dev)
        hm.internal_add_hmatrix_vector_product("N", 1.0, H, xin, 0.0, ref)
        for _ in range(3):
            yg.zero_()
            gp()
        torch.cuda.synchronize()
        assert gp.graph is not None and torch.equal(yg, ref), ("graphed", sym)
        # multi-RHS row-major
        X = torch.randn((n, 4), dtype=tdt, generator=g).to(dev)
        Y0 = torch.randn((n, 4), dtype=tdt, generator=g).to(dev)
        ref = Y0.clone()
        hm.internal_add_hmatrix_matrix_product_row_major("N", alpha, H, X, beta, ref, 4)
        Y = Y0.clone()
        D.internal_add_distributed_operator_matrix_product_row_major_global_to_global("N", alpha, A, X, beta, Y, 4)
        assert rel(Y, ref) < 1e-13, ("rowmajor", sym)
        # block-diagonal operator: with one partition it is the whole operator again
        L = D.DefaultLocalApproximationBuilder(gen, T, T, tb, dtype=dtype)
        y = y0.clone()
        D.internal_add_distributed_operator_vector_product_local_to_local("N", alpha, L.distributed_operator, xin, beta, y)
        ref = y0.clone()
        hm.internal_add_hmatrix_vector_product("N", alpha, H, xin, beta, ref)
        assert rel(y, ref) < 1e-12, ("blockdiag", sym, rel(y, ref))
        print("ok", sym, np.dtype(dtype).name)
    dist.destroy_process_group()
''') % ROOT


def test_distributed_layer_on_gpu_with_rccl():
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("ok ") == 3, out.stdout


MULTI_SCRIPT = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    import htool_amd as hm
    from htool_amd import distributed as D
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)  # every rank on the box's single GPU; gloo carries the collectives (RCCL refuses a shared device)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    rel = lambda a, b: float(torch.linalg.norm(a - b) / torch.linalg.norm(b))
    n = 6001
    x = hm.create_geometry("ball", n)
    b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(60)
    T = b.create_cluster_tree(n, 3, x, 2, world)
    tp = D.PartitionFromCluster(T)
    off, sz = tp.get_offset_of_partition(rank), tp.get_size_of_partition(rank)
    for dtype, tdt, sym, uplo in ((np.float64, torch.float64, "N", "N"), (np.float64, torch.float64, "S", "L"), (np.complex128, torch.complex128, "H", "U")):
        cplx = dtype == np.complex128
        tb = hm.HMatrixTreeBuilder(1e-6, 10.0, sym, uplo)
        tb.set_low_rank_generator("sympartialACA" if sym != "N" else "partialACA")
        gen = hm.InvDistGenerator(3, x, x, 1e-5, 1.0, 1.0, 0.5 if cplx else 0.0, sym == "H")
        B = D.DefaultApproximationBuilder(gen, T, T, tb, dtype=dtype)   # this rank's block rows on the GPU
        A = B.distributed_operator
        Hfull = tb.build(gen, T, T, dtype=dtype)                        # the whole operator, for reference
        g = torch.Generator().manual_seed(0)
        xin = torch.randn(n, dtype=tdt, generator=g).to(dev)
        y0 = torch.randn(n, dtype=tdt, generator=g).to(dev)
        alpha, beta = (1.5 - 0.5j, 0.25 + 1j) if cplx else (1.5, 0.25)
        for trans in ("N",) + (("T",) if sym != "H" else ()) + (("C",) if cplx and sym != "S" else ()):
            ref = y0.clone()
            hm.internal_add_hmatrix_vector_product(trans, alpha, Hfull, xin, beta, ref)
            y = y0.clone()
            D.internal_add_distributed_operator_vector_product_global_to_global(trans, alpha, A, xin, beta, y)
            assert rel(y, ref) < 1e-12, ("g2g", sym, trans, rel(y, ref))
            yl = y0[off:off + sz].clone()
            D.internal_add_distributed_operator_vector_product_local_to_local(trans, alpha, A, xin[off:off + sz].clone(), beta, yl)
            assert rel(yl, ref[off:off + sz]) < 1e-12, ("l2l", sym, trans, rel(yl, ref[off:off + sz]))
            ref_u = y0.clone()
            hm.add_hmatrix_vector_product(trans, alpha, Hfull, xin, beta, ref_u)
            y = y0.clone()
            D.add_distributed_operator_vector_product_global_to_global(trans, alpha, A, xin, beta, y)
            assert rel(y, ref_u) < 1e-12, ("user", sym, trans, rel(y, ref_u))
        if sym == "N":  # multi-RHS row-major and the graphed single-vector product
            X = torch.randn((n, 5), dtype=tdt, generator=g).to(dev)
            Y = torch.zeros((n, 5), dtype=tdt, device=dev)
            D.internal_add_distributed_operator_matrix_product_row_major_global_to_global("N", 1.0, A, X, 0.0, Y, 5)
            R = torch.zeros((n, 5), dtype=tdt, device=dev)
            hm.internal_add_hmatrix_matrix_product_row_major("N", 1.0, Hfull, X, 0.0, R, 5)
            assert rel(Y, R) < 1e-12
            yg = torch.zeros(n, dtype=tdt, device=dev)
            gp = D.GraphedGlobalToGlobalProduct(A, xin, yg)
            gp()
            torch.cuda.synchronize()
            ref = torch.zeros(n, dtype=tdt, device=dev)
            hm.internal_add_hmatrix_vector_product("N", 1.0, Hfull, xin, 0.0, ref)
            assert rel(yg, ref) < 1e-12
    if rank == 0:
        print("multi-rank ok")
    dist.destroy_process_group()
''') % ROOT


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_layer_on_gpu_several_ranks_sharing_the_device(world, tmp_path):
    """The Python DistributedOperator layer with the real HIP engine and MORE than one rank: `world` processes under
    torch.distributed.run share the box's GPU and talk through gloo.  Row-restricted operators per rank (unequal parts for
    world 3), g2g / l2l / user numbering, N / T / C, symmetric and Hermitian storage, multi-RHS and the graphed product against
    the single-process product of the whole operator."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "multi.py"
    script.write_text(MULTI_SCRIPT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "multi-rank ok" in out.stdout


FAMILIES_SCRIPT = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    sys.path.insert(0, os.path.join(%r, "tests"))
    import htool_amd as hm
    from htool_amd import distributed as D
    from helpers import load, MANIFEST
    import test_distributed_gloo as G
    case = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)  # every rank on the box's single GPU; gloo carries the collectives
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    p, g = MANIFEST[case], load(case)
    assert world == p["partitions"]
    x, T, To = G.distprod_cluster_tree(p, world)
    assert np.array_equal(T.get_permutation(), g["perm"])
    cplx = p.get("prec") == "z64"
    sym = p.get("sym", "N")
    tb = hm.HMatrixTreeBuilder(p["eps"], 10.0, sym, p.get("uplo", "N"))
    tb.set_low_rank_generator(p["compressor"])
    gen = hm.InvDistGenerator(3, x, x, 1e-5, 1.0, 1.0, 0.5 if cplx else 0.0, sym == "H")
    dtype = np.complex128 if cplx else np.float64
    block_diagonal = bool(p.get("local", 0))
    B = (D.DefaultLocalApproximationBuilder if block_diagonal else D.DefaultApproximationBuilder)(gen, T, T, tb, dtype=dtype)
    lt = np.asarray(B.hmatrix.leaf_table())
    assert np.array_equal(lt[:, :5], g["r%%d_leaves" %% rank]), "this rank's blocks / ranks differ from htool's under MPI"
    # hmx_dist_*: the same products as one C call each, collectives host-staged over gloo; the block-diagonal operator is registered
    # as a local-to-local operator of a DistributedOperator without global-to-local operator (hmx_dist_add_local_to_local_operator)
    comm = D.NativeCommunicator(backend="gloo")
    native = D.NativeDistributedOperator(None, T, T, comm, block_diagonal_hmatrix=B.hmatrix) if block_diagonal else D.NativeDistributedOperator(B.hmatrix, T, T, comm)
    errs = G.run_distprod_families(B.distributed_operator, rank, world, g, p, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev), native=native)
    bad = sorted((k, float(v)) for k, v in errs.items() if not v < 1e-10)
    assert not bad, bad
    assert native is None or sum(k.startswith("native_") for k in errs) >= 12, sorted(errs)
    if rank == 0:
        print("families ok", len(errs), "max", max(errs.values()))
    dist.destroy_process_group()
''') % (ROOT, ROOT)


@pytest.mark.parametrize("case", ["distprod_ellipse_n2000_p2", "distprod_ellipse_n2400_p4", "distprod_ball_n1500_p2_symL", "distprod_ball_n1500_c3_p3",
                                  "distprod_ellipse_n2000_p2_blockdiag", "distprod_ball_n1200_p2_z64_hermL"])
def test_every_product_family_on_the_gpu_matches_the_reference_under_mpi(case, tmp_path):
    """SURVEY.md 8(f)1 on the HIP engine: `world` processes share the box's GPU (gloo carries the collectives), every rank holds
    its block rows on the device; every product family of distributed_operator/linalg/ -- through the torch.distributed layer AND
    through the C-level hmx_dist_* entry points (row-major local-to-local multi-RHS = what HPDDMOperator::GMV calls, the column-major
    front ends in both numberings) -- against the outputs htool itself produced under mpiexec -n world (tests/golden/distprod_*)."""
    import socket
    from helpers import MANIFEST
    world = MANIFEST[case]["partitions"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "families.py"
    script.write_text(FAMILIES_SCRIPT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), case]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "families ok" in out.stdout
