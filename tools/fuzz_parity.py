"""Randomised parity sweep: the device engine against the CPU oracle on random configurations (geometry, size, leaf size, children,
partitions, eta, eps, compressor, symmetry, coefficient type, row partition, minimal depth).  usage: fuzz_parity.py [seconds] [seed] [max points]"""
import sys
import time

import numpy as np

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import htool_amd as hm
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
STRAT = {"pca_regular": ("largest_extent", "regular", False), "pca_geometric": ("largest_extent", "geometric", False),
         "bbox_regular": ("bounding_box", "regular", False), "bbox_geometric": ("bounding_box", "geometric", False),
         "n_pca_regular": ("largest_extent", "regular", True), "n_bbox_regular": ("bounding_box", "regular", True)}
rel = lambda a, b: np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)
if os.environ.get("FUZZ_RESERVE_GB"):  # every large device array of the library out of one slab (first fit, coalescing free list)
    assert hm.lib().hmx_device_reserve(0, int(float(os.environ["FUZZ_RESERVE_GB"]) * (1 << 30))) == 0
t0, done, worst = time.time(), 0, 0.0
while time.time() - t0 < budget:
    geom = rng.choice(["ellipse", "disk", "ball", "disk2d"])
    n = int(rng.integers(40, NMAX))
    leaf = int(rng.integers(5, 120))
    children = int(rng.choice([2, 2, 2, 3, 4]))
    parts = int(rng.choice([1, 2, 2, 3, 4]))
    strat = rng.choice(list(STRAT))
    eta = float(rng.choice([0.5, 3.0, 10.0, 100.0]))
    eps = float(rng.choice([1e-2, 1e-4, 1e-7, 1e-11]))
    prec = rng.choice(["f64", "f64", "f32", "z64", "c32"])
    cplx = prec in ("z64", "c32")
    sym = rng.choice(["N", "N", "S"] + (["H"] if cplx else []))
    uplo = "N" if sym == "N" else rng.choice(["L", "U"])
    comp = rng.choice(["partialACA", "sympartialACA", "fullACA"]) if sym == "N" else "sympartialACA"
    rank = int(rng.integers(-1, parts)) if rng.random() < 0.4 else -1
    mind = int(rng.choice([0, 0, 1, 2]))
    cfg = dict(geom=geom, n=n, leaf=leaf, children=children, parts=parts, strat=strat, eta=eta, eps=eps, prec=prec, sym=sym, uplo=uplo, comp=comp, rank=rank, mind=mind)
    if os.environ.get("FUZZ_VERBOSE"):
        print(cfg, flush=True)
    dim = 2 if geom == "disk2d" else 3
    x = hm.create_geometry(geom, n)
    b = hm.ClusterTreeBuilder()
    b.set_maximal_leaf_size(leaf)
    b.set_partitioning_strategy(*STRAT[strat])
    try:
        T = b.create_cluster_tree(n, dim, x, children, parts)
    except hm.HmxError:
        continue
    To = O.ClusterTree(x, leaf, children, parts, strat)
    assert np.array_equal(T.get_permutation(), To.perm), ("perm", cfg)
    if len(T.get_clusters_on_partition()) != parts:
        continue
    # a rectangular operator now and then: another point set and its own cluster tree on the source side
    xs, S, So = x, T, To
    if sym == "N" and rank == -1 and rng.random() < 0.25:
        ns = int(rng.integers(40, NMAX))
        xs = hm.create_geometry(str(rng.choice(["disk2d"] if dim == 2 else ["ellipse", "disk", "ball"])), ns) + 0.25
        leaf_s = int(rng.integers(5, 120))
        bs = hm.ClusterTreeBuilder()
        bs.set_maximal_leaf_size(leaf_s)
        S = bs.create_cluster_tree(ns, dim, xs, 2, 1)
        So = O.ClusterTree(xs, leaf_s, 2, 1, "pca_regular")
        assert np.array_equal(S.get_permutation(), So.perm), ("source perm", cfg)
        cfg.update(ns=ns, leaf_s=leaf_s)
    tb = hm.HMatrixTreeBuilder(eps, eta, sym, uplo)
    tb.set_low_rank_generator(comp)
    tb.set_minimal_target_depth(mind)
    tb.set_minimal_source_depth(mind)
    dt = {"f64": np.float64, "f32": np.float32, "z64": np.complex128, "c32": np.complex64}[prec]
    cre, cim = (0.7, -0.4) if cplx else (1.0, 0.0)
    kern = "invdist" if sym == "H" else str(rng.choice(["invdist", "invdist", "helmholtz", "laplace"]))  # device kernel family (include/hmx.h hmx_kernel)
    wk = float(rng.choice([0.5, 3.0, 12.0]))
    cfg.update(kernel=kern, wavenumber=wk)
    gen = {"invdist": lambda: hm.InvDistGenerator(dim, x, xs, 1e-5, 1.0, cre, cim, sym == "H"), "helmholtz": lambda: hm.HelmholtzGenerator(dim, x, xs, wk, 1e-5, 1.0),
           "laplace": lambda: hm.LaplaceGenerator(dim, x, xs, 1e-5, cre, cim)}[kern]()
    H = tb.build(gen, T, S, rank, rank, dtype=dt)
    if rng.random() < 0.2 and comp != "fullACA" and kern == "invdist":  # the same operator through the host-generator route (compiled VirtualGenerator on 1 / 3 / all threads): bit for bit
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import native_inv_dist_generator
        thr = int(rng.choice([1, 3, 0]))
        Hh = tb.build(native_inv_dist_generator(x, xs, 1e-5, 1.0, cre, cim, sym == "H", dtype=dt, threads=thr), T, S, rank, rank, dtype=dt)
        assert np.array_equal(Hh.leaf_table(), H.leaf_table()), ("host generator: structure / ranks", cfg, thr)
        xv = (rng.standard_normal(H.nb_cols()) + (1j * rng.standard_normal(H.nb_cols()) if cplx else 0)).astype(dt)
        ya, yb = np.zeros(H.nb_rows(), dtype=dt), np.zeros(H.nb_rows(), dtype=dt)
        hm.internal_add_hmatrix_vector_product("N", 1.0, H, xv, 0.0, ya)
        hm.internal_add_hmatrix_vector_product("N", 1.0, Hh, xv, 0.0, yb)
        assert np.array_equal(ya, yb), ("host generator: product", cfg, thr)
        del Hh
    if cplx:
        Ho = O.ZHMatrix(To, So, delta=1e-5, scale=1.0, cre=cre, cim=cim, eps=eps, eta=eta, sym=sym, uplo=uplo, compressor=comp, mindepth=mind, rank=rank, c32=prec == "c32", kernel=kern, wavenumber=wk)
    else:
        Ho = O.HMatrix(To, So, delta=1e-5, scale=1.0, eps=eps, eta=eta, sym=sym, uplo=uplo, compressor=comp, mindepth=mind, rank=rank, f32=prec == "f32", kernel=kern, wavenumber=wk)
    lt = H.leaf_table()
    assert np.array_equal(lt[:, :4], Ho.leaves[:, :4]) and np.array_equal(lt[:, 5], Ho.leaves[:, 5]), ("structure", cfg)
    if os.environ.get("FUZZ_ROUNDTRIP") and rng.random() < 0.5:  # binary dump and reload: the products below then run on the reloaded operator
        path = "/tmp/fuzz_%d.hmx" % os.getpid()
        H.save(path)
        H2 = tb.load(path, T, S, rank, rank)
        assert np.array_equal(H2.leaf_table(), lt), ("reload: structure / ranks", cfg)
        os.remove(path)
        H = H2
    no_view = os.environ.get("HMX_TRANS_STREAMS") == "0"
    if os.environ.get("FUZZ_RELEASE") and rng.random() < 0.5:  # only the streams remain
        with_t = bool(rng.integers(0, 2))
        H.release_factors(with_t)
        no_view = no_view or not with_t
    single = prec in ("f32", "c32")
    if not single:
        assert np.array_equal(lt[:, 4], Ho.leaves[:, 4]), ("ranks", cfg, int((lt[:, 4] != Ho.leaves[:, 4]).sum()))
    nr, nc = H.nb_rows(), H.nb_cols()
    tol = (2e-3 if eps < 1e-5 else 5e-3) if single else 1e-9
    if single and not np.array_equal(lt[:, 4], Ho.leaves[:, 4]):
        tol = max(tol, 30 * eps)
    alpha, beta = (1.5 - 0.5j, 0.25 + 1j) if cplx else (1.5, 0.25)
    transes = ["N"] + (["T"] if sym != "H" else []) + (["C"] if cplx and sym != "S" else [])
    if sym != "N" and rank >= 0 and parts > 1 and no_view and bool(np.asarray(lt)[:, 5].any()):  # (column 5: the leaf is in leaves_for_symmetry)
        # a row-restricted symmetric / Hermitian operator (mirrored leaves among ordinary ones) multiplies transposed on its transposed VIEW only;
        # with the view forbidden or impossible (factors released without it) the product is an error, not a slow path
        for trans in transes[1:]:
            try:
                hm.internal_add_hmatrix_vector_product(trans, alpha, H, np.zeros(nr, dtype=dt), beta, np.zeros(nc, dtype=dt))
                raise AssertionError(("transposed product without its view did not fail", trans, cfg))
            except hm.HmxError as e:
                assert "transposed stream layout" in str(e), (str(e), cfg)
        transes = ["N"]
    for trans in transes:
        nin, nout = (nc, nr) if trans == "N" else (nr, nc)
        xin = (rng.standard_normal(nin) + (1j * rng.standard_normal(nin) if cplx else 0)).astype(dt)
        y0 = (rng.standard_normal(nout) + (1j * rng.standard_normal(nout) if cplx else 0)).astype(dt)
        y = y0.copy()
        hm.internal_add_hmatrix_vector_product(trans, alpha, H, xin, beta, y)
        big = np.complex128 if cplx else np.float64
        ref = Ho.matvec(xin.astype(big), trans, alpha, beta, y0.astype(big))
        e = rel(y, ref)
        worst = max(worst, e if not single else 0.0)
        assert e < tol, ("matvec", trans, e, cfg)
        mu = int(rng.choice([2, 3, 5, 7, 8, 11, 16, 19]))
        X = (rng.standard_normal((nin, mu)) + (1j * rng.standard_normal((nin, mu)) if cplx else 0)).astype(dt)
        Y0 = (rng.standard_normal((nout, mu)) + (1j * rng.standard_normal((nout, mu)) if cplx else 0)).astype(dt)
        Y = Y0.copy()
        hm.internal_add_hmatrix_matrix_product_row_major(trans, alpha, H, X, beta, Y, mu)
        e = rel(Y, Ho.matmat_row_major(X.astype(big), trans, alpha, beta, Y0.astype(big)))
        assert e < tol, ("matmat", trans, mu, e, cfg)
    if rank == -1 and os.environ.get("FUZZ_USER"):  # user-numbering front ends = the cluster-numbering products of the permuted operands, bit for bit
        perm_t, perm_s = T.get_permutation(), S.get_permutation()
        for trans in transes:
            perm, permo = (perm_s, perm_t) if trans == "N" else (perm_t, perm_s)  # of the input / of the output
            nin, nout = (nc, nr) if trans == "N" else (nr, nc)
            xu = (rng.standard_normal(nin) + (1j * rng.standard_normal(nin) if cplx else 0)).astype(dt)
            y0 = (rng.standard_normal(nout) + (1j * rng.standard_normal(nout) if cplx else 0)).astype(dt)
            yu, yc = y0.copy(), y0[permo].copy()
            hm.add_hmatrix_vector_product(trans, alpha, H, xu, beta, yu)
            hm.internal_add_hmatrix_vector_product(trans, alpha, H, xu[perm].copy(), beta, yc)
            assert np.array_equal(yu[permo], yc), ("user numbering: vector", trans, cfg)
            mu = int(rng.choice([1, 2, 5, 16, 19]))
            Bu = np.asfortranarray((rng.standard_normal((nin, mu)) + (1j * rng.standard_normal((nin, mu)) if cplx else 0)).astype(dt))
            C0 = np.asfortranarray((rng.standard_normal((nout, mu)) + (1j * rng.standard_normal((nout, mu)) if cplx else 0)).astype(dt))
            Cu, Yc = C0.copy(order="F"), np.ascontiguousarray(C0[permo])
            hm.add_hmatrix_matrix_product(trans, alpha, H, Bu, beta, Cu)
            hm.internal_add_hmatrix_matrix_product_row_major(trans, alpha, H, np.ascontiguousarray(Bu[perm]), beta, Yc, mu)
            assert np.array_equal(Cu[permo], Yc), ("user numbering: column-major matrix", trans, mu, cfg)
        ids = rng.choice(len(lt), size=min(len(lt), 40), replace=False)  # bulk download = block by block
        for k, blk in zip(ids, H.get_blocks(ids)) if not os.environ.get("FUZZ_RELEASE") else ():
            one = H.get_block(int(k))
            same = (np.array_equal(blk[0], one[0]) and np.array_equal(blk[1], one[1])) if lt[k, 4] >= 0 else np.array_equal(blk, one)
            assert same, ("bulk download", int(k), cfg)
    done += 1
print("fuzz parity: %d random configurations ok in %.0fs, worst double-precision product error %.2e" % (done, time.time() - t0, worst))
