"""BASELINE.json's own sizes against the REFERENCE itself (north star: "<= 1e-10 relative error vs CPU reference" at N=1e6).

tests/golden/full_*.npz were written by htool (oracle/_ref/ref_driver, `python tests/golden/make_golden.py full`): N=1e5 ball and
ellipse, N=1e6 ellipse with the bench's minimal block depth, the symmetric ('S','L', sympartialACA) N=1e5 operators, the complex
Hermitian N=1e5 operator of the reference's sign-discontinuous test generator (ranks up to 490: the ACA's pool growth and workgroup teams) and BASELINE
config 5's shape (fp32, 'S','L', eps=1e-6, 16 right-hand sides) at N=1e5, and BASELINE configs[3] rank by rank: the N=1e6 operator
row-partitioned over 8 ranks (`full_ellipse_n1000000_p8_rank<k>`: what rank k of htool's DistributedOperator holds and multiplies
before the Allgatherv), each rank's share built and multiplied on the one GPU a box has.  A fixture holds sha256 of the cluster permutation and of
the leaf table's structure columns, the rank of every leaf, and the reference's products at 4096 fixed rows.  Here the engine
builds the same operator on the GPU: permutation and block structure must hash equal, every rank must equal the reference's, and
the sampled products must agree to 1e-10 (fp64; 2e-5 for the fp32 operator -- the fp32 floor SURVEY.md App. D states)."""
import hashlib

import numpy as np
import pytest

import htool_amd as hm
from helpers import MANIFEST, load
from oracle.oracle import hashed_vector, hashed_zvector

pytestmark = pytest.mark.gpu

FULL_CASES = sorted(k for k, v in MANIFEST.items() if v["mode"] == "full")


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


@pytest.mark.parametrize("name", FULL_CASES)
def test_full_size_matches_the_reference(name):
    p, g = MANIFEST[name], load(name)
    n, mu = p["n"], p.get("mu", 2)
    f32 = p.get("prec") == "f32"
    z64 = p.get("prec") == "z64"  # the reference's complex generators: (1 + i sgn) / (delta + r), sgn = sign(x_t[0] - x_s[0]) for 'H'
    sym, uplo = p.get("sym", "N"), p.get("uplo", "N")
    x = hm.create_geometry(p["geom"], n)
    b = hm.ClusterTreeBuilder()
    b.set_maximal_leaf_size(p["leaf"])
    parts, rank = p.get("partitions", 2), p.get("rank", -1)  # rank >= 0: the block rows of partition `rank` (DistributedOperator's local operator)
    T = b.create_cluster_tree(n, 3, x, 2, parts)
    assert np.array_equal(_sha(np.asarray(T.get_permutation(), dtype=np.int32)), g["perm_sha256"]), "cluster permutation differs from the reference's"
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], sym, uplo)
    tb.set_low_rank_generator(p["compressor"])
    tb.set_minimal_target_depth(p.get("mindepth", 0))
    tb.set_minimal_source_depth(p.get("mindepth", 0))
    gen = hm.InvDistGenerator(3, x, x, 1e-5, 1.0, 1.0, 1.0, sym == "H") if z64 else hm.InvDistGenerator(3, x, x, 1e-5, 1.0)
    dt = np.complex128 if z64 else (np.float32 if f32 else np.float64)
    hashed = (lambda n, s: hashed_zvector(n, s).astype(dt)) if z64 else (lambda n, s: hashed_vector(n, s).astype(dt))
    H = tb.build(gen, T, T, rank, rank, dtype=dt)
    nr = H.nb_rows()
    tab = np.asarray(H.leaf_table())
    assert len(tab) == int(g["nleaves"])
    assert np.array_equal(_sha(tab[:, [0, 1, 2, 3, 5]].astype(np.int32)), g["structure_sha256"]), "block structure differs from the reference's"
    ref_ranks = g["ranks"].astype(np.int64)
    diff = tab[:, 4].astype(np.int64) - ref_ranks
    ndiff = int(np.count_nonzero(diff))
    print("%s: %d leaves, %d ranks differ from the reference (max |diff| %d)" % (name, len(tab), ndiff, int(np.abs(diff).max())))
    # fp64 (and complex double): the stopping test sums in another order than BLAS dot, so a rank could move by one when the estimate lands
    # within rounding of epsilon -- it never does: not a single leaf may differ.
    # fp32 at eps = 1e-6 (configs[4]) is another matter, and the criterion is on the APPROXIMATION, not on a fitted rank distance.  After q
    # rank-1 updates in 24-bit arithmetic a residual entry carries rounding noise of about q u |A| (u = 2^-24), i.e. q u / eps ~ 1 relative to
    # a residual that has decayed to eps |A|: the stopping test sqrt(aux / frob) <= eps runs at the noise floor of the arithmetic, and the
    # iteration at which it first dips below eps is decided by rounding -- in htool (MKL sdot) as here (tree sums).  What must hold instead:
    #   (a) where the ranks differ, the device's factors truncated to the SMALLER of the two ranks already meet the accuracy target up to the
    #       estimator's own slack (ACA's estimate bounds the last correction, not the remaining error: a few eps) -- the extra iterations of
    #       either side are noise-floor iterations, no accuracy is lost or gained (tools/f32_rank_study.py: median 1.0 eps, largest 4.6 eps);
    #   (b) no bias: the sums of all ranks agree to 0.5 % (observed 0.04 %);
    #   (c) the distance decays geometrically, as a stopping time at a noise floor does (observed ratio 0.3 per step: 69 / 25 / 5 / 1 %
    #       at distance 1 / 2 / 3 / >= 4): at most a tenth of the differing leaves beyond distance 2, none beyond 12 (p < 1e-6 per leaf).
    if f32:
        ad = np.abs(diff)
        assert abs(float(diff.sum())) <= 0.005 * float(ref_ranks[ref_ranks > 0].sum())
        assert ndiff == 0 or (float((ad >= 3).sum()) <= 0.10 * ndiff and ad.max() <= 12), np.bincount(ad)
        cand = [i for i in np.nonzero(diff)[0] if int(tab[i, 1]) * int(tab[i, 3]) <= 2000000]
        pick = np.random.default_rng(0).choice(cand, size=min(40, len(cand)), replace=False) if cand else []
        perm = np.asarray(T.get_permutation())
        xs = x.reshape(n, 3)
        worst = 0.0
        for i, (U, V) in zip(pick, H.get_blocks(pick) if len(pick) else []):
            t0, m, s0, nn = (int(v) for v in tab[i, :4])
            P, Q = xs[perm[t0:t0 + m]], xs[perm[s0:s0 + nn]]
            A = 1.0 / (1e-5 + np.sqrt(((P[:, None, :] - Q[None, :, :]) ** 2).sum(-1)))
            r = int(min(tab[i, 4], ref_ranks[i]))
            worst = max(worst, float(np.linalg.norm(A - U[:, :r].astype(np.float64) @ V[:r, :].astype(np.float64)) / np.linalg.norm(A)))
        print("%s: true relative error of the device's blocks truncated to min(rank, reference rank) on %d differing leaves: at most %.2e (eps %g)" % (name, len(pick), worst, p["eps"]))
        # guard, not model: ACA's estimate bounds the last correction, not the remaining error -- "a few eps" is all the theory gives.  Observed over the ten fp32
        # full-size fixtures (deterministic: fixed sample, bit-reproducible device results): 4.6 eps (N = 1e5, one vector) ... 7.9 eps (N = 4e6, rank 4 of 8).  Round 6
        # tried 6 eps on the strength of the first figure alone and failed two fixtures
        assert worst <= 10 * p["eps"], worst
    else:
        assert ndiff == 0
    rows = g["rows"]
    tol = 2e-5 if f32 else 1e-10
    xin = hashed(n, 1)

    def err(a, ref):
        if z64 and not np.iscomplexobj(ref):  # stored as (re, im) pairs
            ref = np.ascontiguousarray(ref).view(np.complex128).reshape(ref.shape[:-1])
        return float(np.linalg.norm(a.astype(np.complex128 if z64 else np.float64) - ref) / np.linalg.norm(ref))

    y = np.zeros(nr, dtype=dt)
    hm.internal_add_hmatrix_vector_product("N", 1.0, H, xin, 0.0, y)
    e1 = err(y[rows], g["yN_a1b0"])
    ab = g["alphabeta"]
    al, be = (complex(ab[0], ab[2]), complex(ab[1], ab[3])) if z64 else (float(ab[0]), float(ab[1]))
    y = hashed(nr, 3)
    hm.internal_add_hmatrix_vector_product("N", al, H, xin, be, y)
    e2 = err(y[rows], g["yN"])
    X = hashed(n * mu, 5).reshape(n, mu)
    Y = hashed(nr * mu, 6).reshape(nr, mu)
    hm.internal_add_hmatrix_matrix_product_row_major("N", al, H, X, be, Y, mu)
    e3 = err(Y[rows], g["YNrm"])
    print("%s: relative error vs the reference at %d rows: %.2e (alpha=1, beta=0), %.2e (alpha=%s, beta=%s), %.2e (%d right-hand sides)" % (name, len(rows), e1, e2, al, be, e3, mu))
    if ndiff == 0:
        assert e1 < tol and e2 < tol and e3 < tol
    else:  # some fp32 ranks moved by one: the products differ by what one cross of those blocks carries (below epsilon)
        assert e1 < 10 * p["eps"] and e2 < 10 * p["eps"] and e3 < 10 * p["eps"]
