"""Complex coefficients on the device: htool's HMatrix<std::complex<double>> / <std::complex<float>> with
complex symmetric ('S') and Hermitian ('H') storage and trans = 'N', 'T', 'C', against the fixtures the reference itself wrote
(tests/golden/*z64*, *c32*) and against the CPU oracle on other sizes.  Through the C ABI (hmx_hmatrix_*_z / *_c).

Bars: structure identical; ranks identical for complex double (a few +-1..2 for complex float, whose ACA stopping test sits
on fp32 rounding); dense entries bit-exact; U V to 1e-9; products to 1e-10 against the reference's own results.
"""
import numpy as np
import pytest

import htool_amd as hm
from helpers import Z_CASES, device_generator, load, params, rel_err
from test_host_structure import build_trees

pytestmark = pytest.mark.gpu


ENGINE_OPTIONS = {}  # options every operator of build_zengine gets (tests that re-run other tests on another code path set it)


def build_zengine(p, compress=True, generator=True, dtype=None, options=None):
    T, S = build_trees(p)
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], p["sym"], p["uplo"], p["reqrank"])
    for k, v in dict(ENGINE_OPTIONS, **(options or {})).items():
        tb.set_option(k, v)
    tb.set_low_rank_generator(p["compressor"])
    gen = device_generator(p, T, S) if generator else None
    dt = dtype or (np.complex64 if p["prec"] == "c32" else np.complex128)
    H = tb.build(gen, T, S, p["rank"], p["rank"], compress=compress, dtype=dt)
    if compress and p["recompress"]:
        H.recompress()
    return T, S, H


def svd_like(p):
    return p["compressor"] == "SVD" or bool(p["recompress"])  # Jacobi vs LAPACK: same truncation rule, +-1 at the threshold


def zinputs(H, g):
    from oracle.oracle import hashed_zvector
    nr, nc = H.nb_rows(), H.nb_cols()
    f = lambda n, s: hashed_zvector(n, s).astype(H.dtype)
    alpha, beta = complex(g["alphabeta"][0], g["alphabeta"][2]), complex(g["alphabeta"][1], g["alphabeta"][3])
    return f(nc, 1), f(nr, 2), f(nr, 3), f(nc, 4), alpha, beta


@pytest.mark.parametrize("name", Z_CASES)
def test_complex_compression_matches_reference(name):
    p, g = params(name), load(name)
    T, S, H = build_zengine(p)
    lt = H.leaf_table()
    assert np.array_equal(lt[:, :4], g["leaves"][:, :4]) and np.array_equal(lt[:, 5], g["leaves"][:, 5])
    assert np.array_equal(lt[:, 4] < 0, g["leaves"][:, 4] < 0)  # the same blocks fall back to dense
    if p["prec"] == "c32" or svd_like(p):
        assert np.abs(lt[:, 4] - g["leaves"][:, 4]).max() <= 2 and (lt[:, 4] != g["leaves"][:, 4]).mean() < 0.05
    else:
        assert np.array_equal(lt[:, 4], g["leaves"][:, 4])
    assert H.stats()["n_false_positive"] == g["rootinfo"][4]
    assert (H.get_symmetry_for_leaves(), H.get_UPLO_for_leaves()) == (chr(g["rootinfo"][5]), chr(g["rootinfo"][6]))
    ptol = 2e-5 if p["prec"] == "c32" else 1e-9
    for k in g:
        if k.startswith("D_"):
            assert np.array_equal(H.get_block(int(k[2:])), g[k].T.astype(H.dtype))  # generator entries bit-exact
        if k.startswith("U_"):
            b = int(k[2:])
            U, V = H.get_block(b)
            if U.shape[1] == g[k].shape[0]:
                assert rel_err(U.astype(np.complex128) @ V.astype(np.complex128), g[k].T @ g["V_%d" % b].T) < ptol


@pytest.mark.parametrize("name", Z_CASES)
def test_complex_workgroup_teams_build_the_same_operator(name, monkeypatch):
    """The team kernels of the ACA (tests/test_gpu_parity.py::test_workgroup_teams_build_the_same_operator) for complex coefficients."""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "ENGINE_OPTIONS", dict(aca_team_min=48, aca_team_after=2, aca_team_slice=64, pool_rank_guess=3))
    test_complex_compression_matches_reference(name)
    test_complex_products_match_reference(name)


@pytest.mark.parametrize("name", Z_CASES)
def test_complex_products_match_reference(name):
    p, g = params(name), load(name)
    T, S, H = build_zengine(p)
    x, xT, y0, y0T, alpha, beta = zinputs(H, g)
    tol = 1e-5 if p["prec"] == "c32" else 1e-10
    if svd_like(p):
        tol = 5e-4  # a rank differing by one at the truncation threshold changes the product by O(eps)
    y = y0.copy()
    hm.internal_add_hmatrix_vector_product("N", alpha, H, x, beta, y)
    assert rel_err(y, g["yN"]) < tol
    for trans, key in (("T", "yT"), ("C", "yC")):
        if key in g:
            y = y0T.copy()
            hm.internal_add_hmatrix_vector_product(trans, alpha, H, xT, beta, y)
            assert rel_err(y, g[key]) < tol
        else:  # add_hmatrix_vector_product.hpp:59-62: 'T' with 'H' leaves, 'C' with 'S' leaves
            with pytest.raises(hm.HmxError, match="not supported"):
                hm.internal_add_hmatrix_vector_product(trans, alpha, H, xT, beta, y0T.copy())
    if "yN_user" in g:
        y = y0.copy()
        hm.add_hmatrix_vector_product("N", alpha, H, x, beta, y)
        assert rel_err(y, g["yN_user"]) < tol
    from oracle.oracle import hashed_zvector
    nr, nc = H.nb_rows(), H.nb_cols()
    X = hashed_zvector(nc * 2, 5).reshape(nc, 2).astype(H.dtype)
    Y = hashed_zvector(nr * 2, 6).reshape(nr, 2).astype(H.dtype)
    hm.internal_add_hmatrix_matrix_product_row_major("N", alpha, H, X, beta, Y, 2)
    assert rel_err(Y, g["YNrm"]) < tol


@pytest.mark.parametrize("sym,uplo,trans_list", [("N", "N", "NTC"), ("S", "L", "NT"), ("H", "U", "NC"), ("H", "L", "NC")])
@pytest.mark.parametrize("mu", [1, 5, 16])
def test_complex_against_oracle_other_size(sym, uplo, trans_list, mu):
    """A size and shape no fixture holds (N = 5000 ellipse, leaf 64), complex double: engine vs CPU oracle, every supported
    trans, single and multiple right-hand sides (fused row-major kernels)."""
    from oracle import oracle as O
    n, eps = 5000, 1e-5
    x3 = hm.create_geometry("ellipse", n)
    b = hm.ClusterTreeBuilder()
    b.set_maximal_leaf_size(64)
    T = b.create_cluster_tree(n, 3, x3, 2, 2)
    tb = hm.HMatrixTreeBuilder(eps, 10.0, sym, uplo)
    tb.set_low_rank_generator("sympartialACA" if sym != "N" else "partialACA")
    H = tb.build(hm.InvDistGenerator(3, x3, x3, 1e-5, 1.0, 0.7, -0.4, sym == "H"), T, T, dtype=np.complex128)
    To = O.ClusterTree(x3, 64, 2, 2)
    Ho = O.ZHMatrix(To, To, delta=1e-5, scale=1.0, cre=0.7, cim=-0.4, eps=eps, eta=10.0, sym=sym, uplo=uplo,
                    compressor="sympartialACA" if sym != "N" else "partialACA", parallel=True)
    assert np.array_equal(H.leaf_table(), Ho.leaves)
    rng = np.random.default_rng(1)
    alpha, beta = 1.5 - 0.5j, -0.3 + 0.8j
    for trans in trans_list:
        if mu == 1:
            xin = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            y0 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            y = y0.copy()
            hm.internal_add_hmatrix_vector_product(trans, alpha, H, xin, beta, y)
            assert rel_err(y, Ho.matvec(xin, trans, alpha, beta, y0)) < 1e-11
        else:
            X = rng.standard_normal((n, mu)) + 1j * rng.standard_normal((n, mu))
            Y0 = rng.standard_normal((n, mu)) + 1j * rng.standard_normal((n, mu))
            Y = Y0.copy()
            hm.internal_add_hmatrix_matrix_product_row_major(trans, alpha, H, X, beta, Y, mu)
            assert rel_err(Y, Ho.matmat_row_major(X, trans, alpha, beta, Y0)) < 1e-11


@pytest.mark.parametrize("mu", [11, 16, 21])
@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-11), (np.complex64, 2e-5)])
def test_eight_complex_rhs_on_the_matrix_cores(dtype, tol, mu, monkeypatch):
    """Groups of 8 complex right-hand sides run as two real MFMAs per complex tile (expand_zmfma8s_kernel / reduce_zmfma8s_kernel:
    matrix/linalg/add_matrix_matrix_product_row_major.hpp:49-84,113-139 is the reference's complex gemm).  mu = 11 = 8 + 2 + 1 against the
    CPU oracle on the operator the oracle itself compressed (complex double; complex float against the VALU kernels, HMX_NO_MFMA=1),
    trans N / T / C, alpha / beta complex; and the matrix-core kernels are what ran.  More than 8 right-hand sides: the expand stage runs
    sweeps of up to 16 in both stages (expand_zmfma16s_kernel / reduce_zmfma16s_kernel: 11 = one ragged sweep, 21 = 16 + ragged 5)."""
    from oracle import oracle as O
    n, eps = 4000, 1e-5
    x3 = hm.create_geometry("ball", n)
    b = hm.ClusterTreeBuilder()
    b.set_maximal_leaf_size(80)
    T = b.create_cluster_tree(n, 3, x3, 2, 2)
    tb = hm.HMatrixTreeBuilder(eps, 10.0, "N", "N")
    tb.set_low_rank_generator("partialACA")
    H = tb.build(hm.InvDistGenerator(3, x3, x3, 1e-5, 1.0, 0.7, -0.4, False), T, T, dtype=dtype)
    rng = np.random.default_rng(3)
    alpha, beta = 1.5 - 0.5j, -0.3 + 0.8j
    Ho = None
    if dtype == np.complex128:
        To = O.ClusterTree(x3, 80, 2, 2)
        Ho = O.ZHMatrix(To, To, delta=1e-5, scale=1.0, cre=0.7, cim=-0.4, eps=eps, eta=10.0, compressor="partialACA", parallel=True)
        assert np.array_equal(H.leaf_table(), Ho.leaves)
    for trans in "NTC":
        X = (rng.standard_normal((n, mu)) + 1j * rng.standard_normal((n, mu))).astype(dtype)
        Y0 = (rng.standard_normal((n, mu)) + 1j * rng.standard_normal((n, mu))).astype(dtype)
        H.set_profiling(True)
        Y = Y0.copy()
        hm.internal_add_hmatrix_matrix_product_row_major(trans, alpha, H, X, beta, Y, mu)
        names = [k for k, _ in H.last_kernel_times()]
        H.set_profiling(False)
        assert any("reduce_zmfma16s" in k for k in names) and any("expand_zmfma16s" in k for k in names), names
        assert (mu > 16) == any("zmfma8s" in k for k in names), names  # 21 = 16 + a ragged sweep of 8
        if Ho is not None:
            ref = Ho.matmat_row_major(X, trans, alpha, beta, Y0)
        else:
            H.set_option("matrix_cores", 0)
            ref = Y0.copy()
            hm.internal_add_hmatrix_matrix_product_row_major(trans, alpha, H, X, beta, ref, mu)
            H.set_option("matrix_cores", 1)
        assert rel_err(Y, ref) < tol, (trans, rel_err(Y, ref))


@pytest.mark.parametrize("name", ["ball_n2000_z64_hermU", "ellipse_n3000_z64_symL", "ball_n2000_z64_p2_rank1"])
def test_complex_upload_download_roundtrip(name, tmp_path):
    """Blocks compressed by the CPU oracle, uploaded through hmx_hmatrix_set_block_*_z, multiplied on the device: equal to
    the CPU leaf loop on the same blocks; get_block returns them bit for bit; save/load keeps the operator."""
    from oracle.oracle import hashed_zvector
    from test_oracle_vs_golden import build_zoracle
    p, To, So, Ho = build_zoracle(name)
    T, S, H = build_zengine(p, compress=False, generator=False)
    assert np.array_equal(H.leaf_table()[:, :4], Ho.leaves[:, :4])
    for b in range(len(Ho.leaves)):
        blk = Ho.block(b)
        if Ho.leaves[b, 4] >= 0:
            H.set_block_lowrank(b, blk[0], blk[1])
        else:
            H.set_block_dense(b, blk)
    H.finalize()
    nr, nc = H.nb_rows(), H.nb_cols()
    alpha, beta = 3.0 + 0.5j, 2.0 - 0.25j
    for trans in ("N",) + (("T",) if p["sym"] != "H" else ()) + (("C",) if p["sym"] != "S" else ()):
        nin, nout = (nc, nr) if trans == "N" else (nr, nc)
        xin, yin = hashed_zvector(nin, 7), hashed_zvector(nout, 8)
        y = yin.copy()
        hm.internal_add_hmatrix_vector_product(trans, alpha, H, xin, beta, y)
        assert rel_err(y, Ho.matvec(xin, trans, alpha, beta, yin)) < 1e-12
    for b in (0, len(Ho.leaves) // 2, len(Ho.leaves) - 1):
        ref, got = Ho.block(b), H.get_block(b)
        if Ho.leaves[b, 4] >= 0:
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        else:
            assert np.array_equal(got, ref)
    H.save(tmp_path / "z.hmx")
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], p["sym"], p["uplo"])
    G = tb.load(tmp_path / "z.hmx", T, S, p["rank"], p["rank"])
    assert G.dtype == np.complex128 and np.array_equal(G.ranks, H.ranks)
    xin = hashed_zvector(nc, 9)
    y1, y2 = np.zeros(nr, dtype=np.complex128), np.zeros(nr, dtype=np.complex128)
    hm.internal_add_hmatrix_vector_product("N", 1.0, H, xin, 0.0, y1)
    hm.internal_add_hmatrix_vector_product("N", 1.0, G, xin, 0.0, y2)
    assert rel_err(y2, y1) < 1e-14


class _HostComplex(hm.VirtualGenerator):
    def __init__(self, xt, xs, delta, scale, cre, cim, herm):
        self.xt, self.xs, self.p = xt, xs, (delta, scale, cre, cim, herm)

    def copy_submatrix(self, M, N, rows, cols):
        delta, scale, cre, cim, herm = self.p
        d = self.xt[rows][:, None, :] - self.xs[cols][None, :, :]
        s = np.zeros((M, N))
        for q in range(d.shape[2]):  # the generator's own summation order
            s = s + d[:, :, q] * d[:, :, q]
        den = delta + scale * np.sqrt(s)
        sgn = np.sign(d[:, :, 0]) if herm else 1.0
        return cre / den + 1j * ((cim * sgn) / den)


@pytest.mark.parametrize("name", ["ball_n2000_z64_hermU", "ball_n2000_z64_partial", "ball_n1200_z64_fullACA"])
def test_complex_host_callback_generator(name):
    """hmx_hmatrix_set_callback_z: the user's complex generator on the host, lock-step ACA on the device."""
    p, g = params(name), load(name)
    T, S = build_trees(p)
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], p["sym"], p["uplo"], p["reqrank"])
    tb.set_low_rank_generator(p["compressor"])
    A = _HostComplex(T.coordinates, S.coordinates, p["delta"], p["scale"], p["cre"], p["cim"], p["sym"] == "H")
    H = tb.build(A, T, S, p["rank"], p["rank"], dtype=np.complex128)
    assert np.array_equal(H.leaf_table(), g["leaves"])
    x, xT, y0, y0T, alpha, beta = zinputs(H, g)
    y = y0.copy()
    hm.internal_add_hmatrix_vector_product("N", alpha, H, x, beta, y)
    assert rel_err(y, g["yN"]) < 1e-10


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_complex_svd_and_recompression_against_oracle(dtype):
    """SVD compressor and SVD recompression for complex coefficients (complex one-sided Jacobi): against the CPU oracle's
    restatement on a case no fixture holds; Eckart-Young check of the fixed-rank SVD on one block."""
    from oracle import oracle as O
    n, eps = 1500, 1e-5 if dtype == np.complex128 else 1e-3
    x3 = hm.create_geometry("ball", n)
    b = hm.ClusterTreeBuilder()
    b.set_maximal_leaf_size(40)
    T = b.create_cluster_tree(n, 3, x3, 2, 2)
    To = O.ClusterTree(x3, 40, 2, 2)
    gen = hm.InvDistGenerator(3, x3, x3, 1e-5, 1.0, 0.6, 0.9)
    for comp, recompress in (("SVD", False), ("partialACA", True)):
        tb = hm.HMatrixTreeBuilder(eps, 10.0, "N", "N")
        tb.set_low_rank_generator(comp)
        H = tb.build(gen, T, T, dtype=dtype)
        Ho = O.ZHMatrix(To, To, delta=1e-5, scale=1.0, cre=0.6, cim=0.9, eps=eps, eta=10.0, compressor=comp, c32=dtype == np.complex64, parallel=True)
        if recompress:
            before = H.ranks.copy()
            H.recompress()
            Ho.recompress(eps)
            assert (H.ranks <= before).all() and H.ranks[H.ranks >= 0].sum() < before[before >= 0].sum()
        lt = H.leaf_table()
        assert np.array_equal(lt[:, :4], Ho.leaves[:, :4])
        assert np.abs(lt[:, 4] - Ho.leaves[:, 4]).max() <= 2 and (lt[:, 4] != Ho.leaves[:, 4]).mean() < 0.05
        rng = np.random.default_rng(5)
        xin = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(dtype)
        y = np.zeros(n, dtype=dtype)
        hm.internal_add_hmatrix_vector_product("N", 1.0, H, xin, 0.0, y)
        assert rel_err(y, Ho.matvec(xin.astype(np.complex128), "N", 1.0, 0.0)) < 20 * eps
        # accuracy of the compressed operator itself against the dense generator on a row slab
        rows = np.arange(0, 200)
        perm = T.get_permutation()
        d = x3[perm][rows][:, None, :] - x3[perm][None, :, :]
        A = (0.6 + 0.9j) / (1e-5 + np.sqrt((d ** 2).sum(-1)))
        assert rel_err(y[rows], A @ xin.astype(np.complex128)) < 50 * eps


def test_complex_dtype_mismatch_is_refused():
    p = params("ball_n2000_z64_partial")
    T, S = build_trees(p)
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], "N", "N")
    tb.set_low_rank_generator("partialACA")
    H = tb.build(hm.InvDistGenerator(3, T.coordinates, S.coordinates, 1e-5, 1.0, 1.0, 1.0), T, S, dtype=np.complex128)
    with pytest.raises(hm.HmxError, match="complex128"):
        hm.internal_add_hmatrix_vector_product("N", 1.0, H, np.zeros(p["n"]), 0.0, np.zeros(p["n"]))


def test_complex_symmetric_expanded_layout():
    """The default layout of symmetric / Hermitian storage is the stored triangle with the fused product (what every other test here runs).
    Option sym_storage = 1: mirrored leaves laid out explicitly (conjugated for 'H')."""
    for name in ("ellipse_n3000_z64_symL", "ball_n2000_z64_hermU", "ball_n2000_c32_hermL"):
        p, g = params(name), load(name)
        T, S, H = build_zengine(p, options=dict(sym_storage=1))
        x, xT, y0, y0T, alpha, beta = zinputs(H, g)
        tol = 1e-5 if p["prec"] == "c32" else 1e-10
        y = y0.copy()
        hm.internal_add_hmatrix_vector_product("N", alpha, H, x, beta, y)
        assert rel_err(y, g["yN"]) < tol
        key, trans = ("yT", "T") if "yT" in g else ("yC", "C")
        y = y0T.copy()
        hm.internal_add_hmatrix_vector_product(trans, alpha, H, xT, beta, y)
        assert rel_err(y, g[key]) < tol


@pytest.mark.parametrize("name", ["ball_n2000_z64_p2_hermL_rank0", "ball_n2000_z64_hermU", "ball_n2000_z64_partial"])
@pytest.mark.parametrize("layout", [0, 1])
def test_conjugate_transposed_products_without_the_transposed_layout(name, layout):
    """'C' (and 'T' where the reference allows it) with the transposed stream layout forbidden (option transposed_layout = 0): an ordinary operator
    runs on its stored data; a row-restricted Hermitian operator (mirrored leaves among ordinary ones) only runs on its transposed view, where the
    mirrored leaves are applied CONJUGATED -- forbidding the view is reported as an error."""
    p, g = params(name), load(name)
    T, S, H = build_zengine(p)
    H.set_option("transposed_layout", layout)
    x, xT, y0, y0T, alpha, beta = zinputs(H, g)
    for trans, key in (("T", "yT"), ("C", "yC")):
        if key in g:
            y = y0T.copy()
            if layout == 0 and p["sym"] != "N" and p["rank"] >= 0:
                with pytest.raises(hm.HmxError, match="transposed stream layout"):
                    hm.internal_add_hmatrix_vector_product(trans, alpha, H, xT, beta, y)
                continue
            hm.internal_add_hmatrix_vector_product(trans, alpha, H, xT, beta, y)
            assert rel_err(y, g[key]) < 1e-10, (trans, rel_err(y, g[key]))
            X = np.stack([xT, 2 * xT, -xT], axis=1).copy()
            Y = np.stack([y0T, y0T, y0T], axis=1).copy()
            hm.internal_add_hmatrix_matrix_product_row_major(trans, alpha, H, X, beta, Y, 3)
            assert rel_err(Y[:, 0], g[key]) < 1e-10 and rel_err(Y[:, 1] - beta * y0T, 2 * (g[key] - beta * y0T)) < 1e-9


@pytest.mark.parametrize("name", ["ellipse_n3000_z64_symL", "ball_n2000_z64_hermU", "ball_n2000_c32_hermL", "ball_n2000_z64_p2_hermL_rank0"])
def test_complex_stored_triangle_product_with_several_right_hand_sides(name):
    """Several right-hand sides on the STORED TRIANGLE of a complex symmetric / Hermitian operator (expand_sym_mu_kernel / rowsym_mu_kernel: the
    fused product for groups of 8 on the VALU; the reference runs the mirror pass on the same leaves with complex symm / hemm,
    hmatrix/linalg/add_hmatrix_matrix_product_row_major.hpp:100-106,160-170, matrix/linalg/add_matrix_matrix_product_row_major.hpp:113-139):
    2, 3, 8, 11 and 19 right-hand sides (ragged groups of 8) against the fused single-vector product column by column, against the expanded
    view and the reference's own row-major product; no expanded copy is allocated, results are bit-reproducible; a row-restricted Hermitian
    operator (mirrored leaves among ordinary ones) runs the same sweeps."""
    from oracle.oracle import hashed_zvector
    p, g = params(name), load(name)
    T, S, H = build_zengine(p)
    H.set_option("sym_multi_rhs", 1)
    x, xT, y0, y0T, alpha, beta = zinputs(H, g)
    tol = 2e-5 if p["prec"] == "c32" else 1e-12
    nr, nc = H.nb_rows(), H.nb_cols()
    rng = np.random.default_rng(11)
    for mu in (2, 3, 8, 11, 19):
        X = (rng.standard_normal((nc, mu)) + 1j * rng.standard_normal((nc, mu))).astype(H.dtype)
        Y0 = (rng.standard_normal((nr, mu)) - 1j * rng.standard_normal((nr, mu))).astype(H.dtype)
        H.set_profiling(True)
        Y = Y0.copy()
        hm.internal_add_hmatrix_matrix_product_row_major("N", alpha, H, X, beta, Y, mu)
        names = [k for k, _ in H.last_kernel_times()]
        assert any("expand_sym_zmfma8" in k for k in names) and any("rowsym_zmfma8" in k for k in names), names  # groups of 8 on the matrix cores
        H.set_option("matrix_cores", 0)  # ... and the same sweeps on the VALU (groups of 2 / 4 / 8)
        Yv = Y0.copy()
        hm.internal_add_hmatrix_matrix_product_row_major("N", alpha, H, X, beta, Yv, mu)
        names = [k for k, _ in H.last_kernel_times()]
        assert any("expand_sym_mu" in k for k in names) and any("rowsym_mu" in k for k in names), names
        H.set_option("matrix_cores", 1)
        H.set_profiling(False)
        assert rel_err(Yv, Y) < tol, (mu, rel_err(Yv, Y))
        ref = Y0.copy()
        for c in range(mu):
            y = np.ascontiguousarray(Y0[:, c])
            hm.internal_add_hmatrix_vector_product("N", alpha, H, np.ascontiguousarray(X[:, c]), beta, y)
            ref[:, c] = y
        assert rel_err(Y, ref) < tol, (mu, rel_err(Y, ref))
        Y2 = Y0.copy()
        hm.internal_add_hmatrix_matrix_product_row_major("N", alpha, H, X, beta, Y2, mu)
        assert np.array_equal(Y, Y2)  # fixed summation order
    assert H.stats()["expanded_bytes"] == 0
    if p["sym"] == "H" and p["rank"] < 0:  # a square Hermitian operator is its own conjugate transpose: 'C' runs the same sweeps
        Yc = Y0.copy()
        hm.internal_add_hmatrix_matrix_product_row_major("C", alpha, H, X, beta, Yc, mu)
        assert np.array_equal(Yc, Y)
    Xg = hashed_zvector(nc * 2, 5).reshape(nc, 2).astype(H.dtype)
    Yg = hashed_zvector(nr * 2, 6).reshape(nr, 2).astype(H.dtype)
    hm.internal_add_hmatrix_matrix_product_row_major("N", alpha, H, Xg, beta, Yg, 2)
    assert rel_err(Yg, g["YNrm"]) < (1e-5 if p["prec"] == "c32" else 1e-10)  # the reference's own product
    H.set_option("sym_multi_rhs", 0)  # ... and the expanded view gives the same product (another summation order)
    Y3 = Y0.copy()
    hm.internal_add_hmatrix_matrix_product_row_major("N", alpha, H, X, beta, Y3, mu)
    assert H.stats()["expanded_bytes"] > 0 and rel_err(Y3, Y) < tol


@pytest.mark.parametrize("name", [c for c in Z_CASES if params(c)["compressor"] in ("partialACA", "sympartialACA")])
def test_one_wave_kernel_builds_the_same_crosses_complex(name):
    """aca_wave_kernel against aca_kernel, complex coefficients: bit for bit (see test_gpu_parity.check_one_wave_kernel)."""
    from test_gpu_parity import check_one_wave_kernel
    p = params(name)
    check_one_wave_kernel(lambda opts: build_zengine(p, options=opts)[2])
