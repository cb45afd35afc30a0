"""Pin the CPU oracle (oracle/hmx_oracle.cpp) against outputs of the reference itself (tests/golden).

Bit-exact: geometry, permutation, cluster table (ints AND fp64 radius/centre), leaf list incl. ranks.
Tolerance: compressed payloads 1e-9 relative (ACA values go through BLAS axpy in the reference, whose
FMA use is vendor-defined), H-matvec 1e-12 relative.
"""
import numpy as np
import pytest

from helpers import HMAT_CASES, LRMAT_CASES, load, params, rel_err
from oracle import oracle as O


def build_oracle(name):
    p = params(name)
    xt = O.geometry(p["geom"], p["n"])
    gp = O.given_partition(p["given"], p["n"], p["partitions"]) if p["given"] != "none" else None
    T = O.ClusterTree(xt, p["leaf"], p["children"], p["partitions"], p["partitioning"], given_partition=gp, given_local=p["given"] == "local",
                      is_complete=bool(p["complete"]))
    if p["nsrc"]:
        xs = O.geometry(p["sgeom"], p["nsrc"], p["sz"])
        S = O.ClusterTree(xs, p["leaf"], p["children"], p["partitions"], p["partitioning"])
    else:
        S = T
    H = O.HMatrix(T, S, kernel=p["kernel"], wavenumber=p["wavenumber"], delta=p["delta"], scale=p["scale"], eps=p["eps"], eta=p["eta"], sym=p["sym"], uplo=p["uplo"],
                  reqrank=p["reqrank"], compressor=p["compressor"], mindepth=p["mindepth"], rank=p["rank"], consistent=bool(p["consistent"]), root_partition=p["local"])
    if p["recompress"]:
        H.recompress(p["eps"])
    return p, T, S, H


@pytest.mark.parametrize("name", ["ball_n2000_partial", "ellipse_n3000_partial"])
def test_geometry_bit_exact(name):
    p, g = params(name), load(name)
    assert np.array_equal(O.geometry(p["geom"], p["n"]), g["xt"])


@pytest.mark.parametrize("name", HMAT_CASES)
def test_cluster_and_block_tree_bit_exact(name):
    g = load(name)
    p, T, S, H = build_oracle(name)
    assert np.array_equal(T.perm, g["t_perm"])
    assert np.array_equal(T.nodes_int, g["t_nodes_int"])
    assert np.array_equal(T.nodes_real, g["t_nodes_real"])  # fp64 radius / centre, bit for bit
    assert np.array_equal(T.partition, g["t_partition"])
    if p["nsrc"]:
        assert np.array_equal(S.perm, g["s_perm"])
        assert np.array_equal(S.nodes_int, g["s_nodes_int"])
    if p["compressor"] == "SVD":
        # Jacobi vs LAPACK gesvd: same truncation rule, ranks may differ by one at the threshold
        assert np.array_equal(H.leaves[:, :4], g["leaves"][:, :4])
        assert np.abs(H.leaves[:, 4] - g["leaves"][:, 4]).max() <= 1
        assert (H.leaves[:, 4] != g["leaves"][:, 4]).mean() < 0.02
    else:
        assert np.array_equal(H.leaves, g["leaves"])  # structure, ranks (-1 = dense) and mirror flags
    assert np.array_equal(H.rootinfo, g["rootinfo"])


@pytest.mark.parametrize("name", HMAT_CASES)
def test_payload_and_matvec(name):
    g = load(name)
    p, T, S, H = build_oracle(name)
    for k in g:
        if k.startswith("U_"):
            b = int(k[2:])
            U, V = H.block(b)
            if U.shape[1] != g[k].shape[0]:
                continue
            if p["compressor"] == "SVD" or p["recompress"]:  # singular vectors are unique only up to sign
                assert rel_err(U @ V, g[k].T @ g["V_%d" % b].T) < 1e-9
            else:
                # factor-wise: rounding differences (MKL's axpy fuses multiply-add) grow with the number of ACA
                # iterations; the product U V is what the matvec sees
                ftol = 1e-9 if U.shape[1] <= 20 else 1e-6
                assert rel_err(U, g[k].T) < ftol and rel_err(V, g["V_%d" % b].T) < ftol
                assert rel_err(U @ V, g[k].T @ g["V_%d" % b].T) < 1e-9
        if k.startswith("D_"):
            assert np.array_equal(H.block(int(k[2:])), g[k].T)  # kernel entries bit-exact
    nr, nc = H.rootinfo[1], H.rootinfo[3]
    x, xT, y0, y0T = O.hashed_vector(nc, 1), O.hashed_vector(nr, 2), O.hashed_vector(nr, 3), O.hashed_vector(nc, 4)
    alpha, beta = g["alphabeta"][:2]
    tol = 1e-12 if p["compressor"] != "SVD" else 5e-4
    assert rel_err(H.matvec(x, "N", alpha, beta, y0), g["yN"]) < tol
    assert rel_err(H.matvec(xT, "T", alpha, beta, y0T), g["yT"]) < tol
    assert rel_err(H.matvec(x, "N", alpha, beta, y0, policy="omp"), g["yN"]) < tol
    X, Y0 = O.hashed_vector(nc * 2, 5).reshape(nc, 2), O.hashed_vector(nr * 2, 6).reshape(nr, 2)
    assert rel_err(H.matmat_row_major(X, "N", alpha, beta, Y0), g["YNrm"]) < tol
    if "yN_user" in g:  # user-numbering front end (add_hmatrix_vector_product.hpp:173-197)
        perm = T.perm
        yc = H.matvec(x[perm], "N", alpha, beta, y0[perm])
        yu = np.empty_like(yc)
        yu[perm] = yc
        assert rel_err(yu, g["yN_user"]) < tol


@pytest.mark.parametrize("name", LRMAT_CASES)
def test_compressors_on_reference_test_block(name):
    """tests/functional_tests/hmatrix/lrmat/test_lrmat_build.hpp: 500x100 block between two disks."""
    g, p = load(name), params(name)
    nr, nc = 500, 100
    xt, xs = O.geometry("disk", nr, 0.0), O.geometry("disk", nc, p["distance"])
    T = O.ClusterTree(xt, 10, 2, 2)
    S0 = O.ClusterTree(xt[:nc].copy(), 10, 2, 2)  # the reference builds the source tree from xt (quirk)
    assert np.array_equal(T.perm, g["t_perm"]) and np.array_equal(S0.perm, g["s_perm"])
    S = O.ClusterTree(xs, 10, 2, 2)
    S.perm[:] = S0.perm  # permutation of the quirky tree, coordinates of the real source points
    import ctypes as C
    O.lib().orc_cluster_destroy(S.h)
    S.h = S0.h
    S0.h = None
    S.coords = xs
    A = O.generate_block(T, S, nr, nc, 0, 0, 0.0, 4 * np.pi)
    for comp in ("partialACA", "sympartialACA", "fullACA", "SVD"):
        rf, Uf, Vf, _, _ = O.compress_block(T, S, comp, nr, nc, 0, 0, 1e-4, reqrank=10)
        ra, Ua, Va, _, sing = O.compress_block(T, S, comp, nr, nc, 0, 0, 1e-4)
        assert [rf, ra, 1] == list(g[comp + "_info"])
        # the reference test's own thresholds (test_lrmat_build.hpp:32-78)
        assert np.linalg.norm(A - Uf @ Vf) < 1e-8
        assert np.linalg.norm(A - Ua @ Va) < 1e-4
        if comp + "_auto_U" in g:
            assert rel_err(Ua @ Va, g[comp + "_auto_U"].T @ g[comp + "_auto_V"].T) < 1e-9
            assert rel_err(Uf @ Vf, g[comp + "_fixed_U"].T @ g[comp + "_fixed_V"].T) < 1e-7
            if comp != "SVD":
                assert rel_err(Ua, g[comp + "_auto_U"].T) < 1e-9
        if comp == "SVD":  # Eckart-Young (test_lrmat_build_SVD.cpp:81-93)
            for k in range(1, 6):
                _, Uk, Vk, _, _ = O.compress_block(T, S, comp, nr, nc, 0, 0, 1e-4, reqrank=k)
                assert abs(np.linalg.norm(A - Uk @ Vk) - np.sqrt((sing[k:] ** 2).sum())) < 1e-10


from helpers import F32_CASES  # noqa: E402


@pytest.mark.parametrize("name", F32_CASES)
def test_fp32_oracle_against_reference(name):
    """HMatrix<float,double>: structure bit-exact (geometry is fp64), dense entries = fp64 entries rounded to float
    (bit-exact), ranks equal, factors / products to float rounding."""
    g, p = load(name), params(name)
    xt = O.geometry(p["geom"], p["n"])
    T = O.ClusterTree(xt, p["leaf"], p["children"], p["partitions"], p["partitioning"])
    H = O.HMatrix(T, T, kernel=p["kernel"], wavenumber=p["wavenumber"], delta=p["delta"], scale=p["scale"], eps=p["eps"], eta=p["eta"], sym=p["sym"], uplo=p["uplo"],
                  compressor=p["compressor"], rank=p["rank"], f32=True)
    assert np.array_equal(H.leaves[:, :4], g["leaves"][:, :4]) and np.array_equal(H.leaves[:, 5], g["leaves"][:, 5])
    if p["eps"] >= 1e-4:
        assert np.array_equal(H.leaves[:, 4], g["leaves"][:, 4])
    else:
        # eps = 1e-6 sits on the fp32 noise floor: the ACA stopping estimate (BLAS sdot / saxpy in the reference, summation
        # order and FMA use vendor-defined) decides +-a few iterations differently; SURVEY.md App. D states the fp32 bar
        # as ~1e-5 relative on the product
        assert np.array_equal(H.leaves[:, 4] < 0, g["leaves"][:, 4] < 0)
        assert np.abs(H.leaves[:, 4] - g["leaves"][:, 4]).max() <= 3
    for k in g:
        if k.startswith("D_"):
            assert np.array_equal(H.block(int(k[2:])), g[k].T)
        if k.startswith("U_"):
            b = int(k[2:])
            U, V = H.block(b)
            assert rel_err(U @ V, g[k].T @ g["V_%d" % b].T) < 2e-5
    nr, nc = H.rootinfo[1], H.rootinfo[3]
    f = lambda n, s: O.hashed_vector(n, s).astype(np.float32).astype(np.float64)
    alpha, beta = g["alphabeta"][:2]
    assert rel_err(H.matvec(f(nc, 1), "N", alpha, beta, f(nr, 3)), g["yN"]) < 1e-5
    assert rel_err(H.matvec(f(nr, 2), "T", alpha, beta, f(nc, 4)), g["yT"]) < 1e-5


from helpers import Z_CASES  # noqa: E402


def build_zoracle(name):
    p = params(name)
    xt = O.geometry(p["geom"], p["n"])
    T = O.ClusterTree(xt, p["leaf"], p["children"], p["partitions"], p["partitioning"])
    S = T
    if p["nsrc"]:
        S = O.ClusterTree(O.geometry(p["sgeom"], p["nsrc"], p["sz"]), p["leaf"], p["children"], p["partitions"], p["partitioning"])
    H = O.ZHMatrix(T, S, kernel=p["kernel"], wavenumber=p["wavenumber"], delta=p["delta"], scale=p["scale"], cre=p["cre"], cim=p["cim"], eps=p["eps"], eta=p["eta"], sym=p["sym"],
                   uplo=p["uplo"], reqrank=p["reqrank"], compressor=p["compressor"], rank=p["rank"], c32=p["prec"] == "c32")
    if p["recompress"]:
        H.recompress(p["eps"])
    return p, T, S, H


@pytest.mark.parametrize("name", Z_CASES)
def test_complex_oracle_against_reference(name):
    """HMatrix<std::complex<double|float>>: complex symmetric ('S') and Hermitian ('H') storage, trans = 'N', 'T', 'C'.
    Structure and ranks equal; dense entries bit-exact; factors / products to rounding (zaxpy / zdot order in the
    reference's BLAS is vendor-defined)."""
    g = load(name)
    p, T, S, H = build_zoracle(name)
    c32 = p["prec"] == "c32"
    assert np.array_equal(H.leaves[:, :4], g["leaves"][:, :4]) and np.array_equal(H.leaves[:, 5], g["leaves"][:, 5])
    assert np.array_equal(H.leaves[:, 4] < 0, g["leaves"][:, 4] < 0)
    svd_like = p["compressor"] == "SVD" or p["recompress"]  # Jacobi vs LAPACK: same truncation rule, +-1 at the threshold
    if c32 or svd_like:
        assert np.abs(H.leaves[:, 4] - g["leaves"][:, 4]).max() <= 2 and (H.leaves[:, 4] != g["leaves"][:, 4]).mean() < 0.05
    else:
        assert np.array_equal(H.leaves[:, 4], g["leaves"][:, 4])
    assert np.array_equal(H.rootinfo, g["rootinfo"])
    ptol, vtol = (2e-5, 1e-5) if c32 else (1e-9, 1e-12)
    if svd_like:
        vtol = max(vtol, 5e-4)  # a rank differing by one at the threshold changes the product by O(eps)
    for k in g:
        if k.startswith("D_"):
            assert np.array_equal(H.block(int(k[2:])), g[k].T)
        if k.startswith("U_"):
            b = int(k[2:])
            U, V = H.block(b)
            if U.shape[1] == g[k].shape[0]:
                assert rel_err(U @ V, g[k].T @ g["V_%d" % b].T) < ptol
    nr, nc = H.rootinfo[1], H.rootinfo[3]
    rnd = (lambda v: v.astype(np.complex64).astype(np.complex128)) if c32 else (lambda v: v)
    x, xT, y0, y0T = (rnd(O.hashed_zvector(n, s)) for n, s in ((nc, 1), (nr, 2), (nr, 3), (nc, 4)))
    alpha, beta = complex(g["alphabeta"][0], g["alphabeta"][2]), complex(g["alphabeta"][1], g["alphabeta"][3])
    assert rel_err(H.matvec(x, "N", alpha, beta, y0), g["yN"]) < vtol
    assert rel_err(H.matvec(x, "N", alpha, beta, y0, policy="omp"), g["yN"]) < vtol
    if "yT" in g:
        assert rel_err(H.matvec(xT, "T", alpha, beta, y0T), g["yT"]) < vtol
    if "yC" in g:
        assert rel_err(H.matvec(xT, "C", alpha, beta, y0T), g["yC"]) < vtol
    X, Y0 = rnd(O.hashed_zvector(nc * 2, 5)).reshape(nc, 2), rnd(O.hashed_zvector(nr * 2, 6)).reshape(nr, 2)
    assert rel_err(H.matmat_row_major(X, "N", alpha, beta, Y0), g["YNrm"]) < vtol
    if "yN_user" in g:
        perm = T.perm
        yc = H.matvec(x[perm], "N", alpha, beta, y0[perm])
        yu = np.empty_like(yc)
        yu[perm] = yc
        assert rel_err(yu, g["yN_user"]) < vtol
