"""The host-generator route on all cores (hmx_hmatrix_set_callback + hmx_hmatrix_set_callback_threads; the reference's
HMatrixTreeBuilder::openmp_compute_blocks, hmatrix/tree_builder/tree_builder.hpp:603-648, calls the user's VirtualGenerator from an
OpenMP parallel for).  The generator is examples/host_generator.c -- compiled code libhmx knows nothing about -- and computes the same
function as the built-in device kernel, so the two routes must give the same operator BIT FOR BIT: same leaf table, same U / V / dense
payloads, same products; and the reference's structure, ranks and products on every fixture."""
import numpy as np
import pytest

import htool_amd as hm
from helpers import device_generator, load, native_inv_dist_generator, params, rel_err
from test_host_structure import build_trees

pytestmark = pytest.mark.gpu

REAL = ["ball_n2000_partial", "ellipse_n3000_symL_default", "ball_n2000_p2_symU_rank1", "rect_ball1500_disk1000", "ball_n1200_fullACA", "ball_n1200_SVD",
        "ball_n1200_reqrank5", "ball_n2000_n_bbox_c8"]
COMPLEX = ["ball_n2000_z64_hermU", "ball_n2000_z64_partial", "ball_n1200_z64_fullACA"]
SINGLE = ["ellipse_n3000_f32_partial", "ellipse_n3000_f32_symL_eps1e-6", "ball_n2000_c32_hermL", "ellipse_n3000_c32_partial"]
NP = {"f64": np.float64, "f32": np.float32, "z64": np.complex128, "c32": np.complex64}


def builder(p):
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], p["sym"], p["uplo"], p["reqrank"])
    tb.set_low_rank_generator(p["compressor"])
    tb.set_minimal_target_depth(p["mindepth"])
    tb.set_minimal_source_depth(p["mindepth"])
    return tb


def both_routes(p, threads, options=None):
    T, S = build_trees(p)
    dt = NP[p["prec"]]
    herm = p["sym"] == "H"
    dev = device_generator(p, T, S)
    host = native_inv_dist_generator(T.coordinates, S.coordinates, p["delta"], p["scale"], p["cre"], p["cim"], herm, dtype=dt, threads=threads)
    bd, bh = builder(p), builder(p)
    for k, v in (options or {}).items():
        bd.set_option(k, v)
        bh.set_option(k, v)
    Hd = bd.build(dev, T, S, p["rank"], p["rank"], dtype=dt)
    Hh = bh.build(host, T, S, p["rank"], p["rank"], dtype=dt)
    return Hd, Hh


def same_operator(Hd, Hh):
    assert np.array_equal(Hh.leaf_table(), Hd.leaf_table())
    for b, r in enumerate(Hd.ranks):
        if r >= 0:
            (U, V), (U2, V2) = Hd.get_block(b), Hh.get_block(b)
            assert np.array_equal(U, U2) and np.array_equal(V, V2), b
        else:
            assert np.array_equal(Hd.get_block(b), Hh.get_block(b)), b


@pytest.mark.parametrize("threads", [1, 0, 7])
@pytest.mark.parametrize("name", REAL + COMPLEX + SINGLE)
def test_compiled_generator_on_all_threads_builds_the_device_operator(name, threads):
    p, g = params(name), load(name)
    Hd, Hh = both_routes(p, threads)
    same_operator(Hd, Hh)
    ref = g["leaves"]
    tab = Hh.leaf_table()
    if p["prec"] in ("f32", "c32"):  # single precision: the two routes agree bit for bit (above); against the reference the structure is exact and
        assert np.array_equal(tab[:, :4], ref[:, :4])  # the ranks are compared where tests/test_gpu_parity.py / test_gpu_complex.py compare them
    elif p["compressor"] == "SVD":
        assert np.array_equal(tab[:, :4], ref[:, :4]) and np.abs(tab[:, 4] - ref[:, 4]).max() <= 1
    else:
        assert np.array_equal(tab, ref)
    rng = np.random.default_rng(3)
    dt = NP[p["prec"]]
    x = rng.standard_normal(Hh.nb_cols()).astype(dt)
    if np.iscomplexobj(x):
        x = x + 1j * rng.standard_normal(Hh.nb_cols()).astype(dt)
    yd, yh = np.zeros(Hh.nb_rows(), dtype=dt), np.zeros(Hh.nb_rows(), dtype=dt)
    hm.internal_add_hmatrix_vector_product("N", 1.0, Hd, x, 0.0, yd)
    hm.internal_add_hmatrix_vector_product("N", 1.0, Hh, x, 0.0, yh)
    assert np.array_equal(yd, yh)  # same streams, same kernels


@pytest.mark.parametrize("name", ["ball_n2000_partial", "ellipse_n3000_symL_default", "ball_n2000_z64_hermU"])
def test_pool_growth_parks_blocks_and_continues_them(name):
    """A pool sized for rank 1 runs out in the first iterations: the blocks are parked with their row pivot, the pool grows and they
    continue -- several times -- to the same operator."""
    p = params(name)
    Hd, Hh = both_routes(p, 5, options=dict(pool_rank_guess=1))
    same_operator(Hd, Hh)
    assert np.array_equal(Hh.leaf_table(), load(name)["leaves"])


def test_batches_larger_than_the_first_slot_buffer():
    """N = 150 000: the first batches hold blocks whose lines add up to more than a slot's first buffer (1 MiB), so the slots grow; the
    last ones hold thousands of leaf-sized blocks.  Against the device-kernel build."""
    n = 150000
    x = hm.create_geometry("ellipse", n)
    b = hm.ClusterTreeBuilder()
    b.set_maximal_leaf_size(100)
    T = b.create_cluster_tree(n, 3, x, 2, 2)

    def build(gen):
        tb = hm.HMatrixTreeBuilder(1e-4, 10.0, "N", "N")
        tb.set_low_rank_generator("partialACA")
        return tb.build(gen, T, T)
    Hd = build(hm.InvDistGenerator(3, x, x, 1e-5, 1.0))
    Hh = build(native_inv_dist_generator(x, x, 1e-5, 1.0))
    assert np.array_equal(Hh.leaf_table(), Hd.leaf_table())
    rng = np.random.default_rng(1)
    xin = rng.standard_normal(n)
    yd, yh = np.zeros(n), np.zeros(n)
    hm.internal_add_hmatrix_vector_product("N", 1.0, Hd, xin, 0.0, yd)
    hm.internal_add_hmatrix_vector_product("N", 1.0, Hh, xin, 0.0, yh)
    assert np.array_equal(yd, yh)


@pytest.mark.parametrize("name", ["ball_n2000_partial", "ellipse_n3000_symL_default", "ball_n2000_z64_hermU", "rect_ball1500_disk1000"])
def test_bulk_download_equals_block_by_block(name):
    """hmx_hmatrix_get_blocks (gathered on the device, a few large copies) against hmx_hmatrix_get_block leaf by leaf: bitwise, whole
    operator and a shuffled subset."""
    p = params(name)
    Hd, _ = both_routes(p, 1)
    one = [Hd.get_block(b) for b in range(len(Hd.ranks))]
    bulk = Hd.get_blocks()
    sel = np.random.default_rng(0).permutation(len(Hd.ranks))[: max(1, len(Hd.ranks) // 3)]
    part = Hd.get_blocks(sel)
    for got, idx in ((bulk, range(len(Hd.ranks))), (part, sel)):
        for blk, b in zip(got, idx):
            if Hd.ranks[b] >= 0:
                assert np.array_equal(blk[0], one[b][0]) and np.array_equal(blk[1], one[b][1]), b
            else:
                assert np.array_equal(blk, one[b]), b


@pytest.mark.parametrize("name", ["full_ellipse_n100000", "full_ball_n100000", "full_ellipse_n1000000", "full_ellipse_n100000_symL"])
def test_host_generator_at_full_size_gives_the_reference_ranks(name):
    """BASELINE's own sizes through the literal drop-in route: the user's generator as compiled host code on all cores, lock-step ACA on the
    device.  Every rank must equal htool's (fixtures written by the reference, tests/test_gpu_full_size_reference.py), the product its product."""
    import hashlib

    from helpers import MANIFEST
    from oracle.oracle import hashed_vector
    p, g = MANIFEST[name], load(name)
    n = p["n"]
    x = hm.create_geometry(p["geom"], n)
    b = hm.ClusterTreeBuilder()
    b.set_maximal_leaf_size(p["leaf"])
    T = b.create_cluster_tree(n, 3, x, 2, p.get("partitions", 2))
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], p.get("sym", "N"), p.get("uplo", "N"))
    tb.set_low_rank_generator(p["compressor"])
    tb.set_minimal_target_depth(p.get("mindepth", 0))
    tb.set_minimal_source_depth(p.get("mindepth", 0))
    H = tb.build(native_inv_dist_generator(x, x, 1e-5, 1.0), T, T, p.get("rank", -1), p.get("rank", -1))
    tab = np.asarray(H.leaf_table())
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(tab[:, [0, 1, 2, 3, 5]].astype(np.int32)).tobytes()).digest(), dtype=np.uint8)
    assert np.array_equal(sha, g["structure_sha256"])
    assert np.array_equal(tab[:, 4].astype(np.int64), g["ranks"].astype(np.int64))
    y = np.zeros(H.nb_rows())
    hm.internal_add_hmatrix_vector_product("N", 1.0, H, hashed_vector(n, 1), 0.0, y)
    assert rel_err(y[g["rows"]], g["yN_a1b0"]) < 1e-10
