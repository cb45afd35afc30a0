"""The reference's on-disk formats either side of the hot path: save_cluster_tree / read_cluster_tree
(clustering/cluster_output.hpp:33-179), save_leaves_with_rank (hmatrix/hmatrix_output.hpp:39-55), matrix_to_bytes /
bytes_to_matrix (matrix/utils/output.hpp:41-75).  The fixtures hold the files the reference itself wrote
(tests/golden/make_golden.py, mode "io"); libhmx must produce the same bytes and read them back the same way."""
import io

import numpy as np
import pytest

import htool_amd as hm
from helpers import MANIFEST, device_generator, load, params
from test_host_structure import build_trees

IO_CASES = sorted(k for k, v in MANIFEST.items() if v["mode"] == "io")


def _bytes(path):
    return np.fromfile(path, dtype=np.uint8)


@pytest.mark.parametrize("name", IO_CASES)
def test_save_cluster_tree_matches_reference_bytes(name, tmp_path):
    p, g = params(name), load(name)
    T, _ = build_trees(p)
    hm.save_cluster_tree(T, tmp_path / "mine")
    assert np.array_equal(_bytes(tmp_path / "mine_cluster_tree.csv"), g["tree"])
    assert np.array_equal(_bytes(tmp_path / "mine_cluster_tree_properties.csv"), g["properties"])


@pytest.mark.parametrize("name", IO_CASES)
def test_read_cluster_tree(name, tmp_path):
    """Reading the reference's files gives the tree htool's read_cluster_tree gives (compared through a second save,
    like the reference's own test, clustering/test_cluster.hpp:115-117) and the integer structure of the built tree."""
    p, g = params(name), load(name)
    g["tree"].tofile(tmp_path / "ref_cluster_tree.csv")
    g["properties"].tofile(tmp_path / "ref_cluster_tree_properties.csv")
    L = hm.read_cluster_tree(tmp_path / "ref_cluster_tree_properties.csv", tmp_path / "ref_cluster_tree.csv")
    hm.save_cluster_tree(L, tmp_path / "again")
    assert np.array_equal(_bytes(tmp_path / "again_cluster_tree.csv"), g["reread_tree"])
    assert np.array_equal(_bytes(tmp_path / "again_cluster_tree_properties.csv"), g["reread_properties"])
    T, _ = build_trees(p)
    assert np.array_equal(L.get_permutation(), T.get_permutation())
    assert np.array_equal(L.nodes_int(), T.nodes_int())
    assert np.array_equal(L.get_clusters_on_partition(), T.get_clusters_on_partition())
    assert (L.get_maximal_depth(), L.get_minimal_depth(), L.get_maximal_leaf_size(), L.is_permutation_local()) == \
           (T.get_maximal_depth(), T.get_minimal_depth(), T.get_maximal_leaf_size(), T.is_permutation_local())
    assert np.allclose(L.nodes_real(), T.nodes_real(), rtol=1e-5, atol=1e-6)  # 6 significant digits survive the file


def test_read_cluster_tree_errors(tmp_path):
    with pytest.raises(hm.HmxError, match="cannot open"):
        hm.read_cluster_tree(tmp_path / "nope_properties.csv", tmp_path / "nope.csv")
    (tmp_path / "bad.csv").write_text("1|2|3\n")
    (tmp_path / "bad_properties.csv").write_text("maximal leaf size: 3\n")
    with pytest.raises(hm.HmxError, match="malformed"):
        hm.read_cluster_tree(tmp_path / "bad_properties.csv", tmp_path / "bad.csv")


def _block_tree(p, T, S):
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], p["sym"], p["uplo"])
    return tb.build_block_tree(T, S, p["rank"], p["rank"])


@pytest.mark.parametrize("name", IO_CASES)
def test_save_leaves_structure(name, tmp_path):
    """Leaf order, relative offsets and format (ranks taken from the reference's file: no GPU here)."""
    p, g = params(name), load(name)
    T, S = build_trees(p)
    bt = _block_tree(p, T, S)
    ref = np.loadtxt(io.BytesIO(g["leaves"].tobytes()), delimiter=",", skiprows=1, dtype=np.int64)
    bt.ranks = ref[:, 4].astype(np.int32)
    hm.save_leaves_with_rank(bt, tmp_path / "leaves")
    assert np.array_equal(_bytes(tmp_path / "leaves.csv"), g["leaves"])


@pytest.mark.parametrize("name", IO_CASES[:2])
def test_matrix_bytes_roundtrip(name, tmp_path):
    g = load(name)
    g["dense0"].tofile(tmp_path / "d.bin")
    D = hm.bytes_to_matrix(tmp_path / "d.bin")
    rows, cols = np.frombuffer(g["dense0"][:8].tobytes(), dtype=np.int32)
    assert D.shape == (rows, cols)
    hm.matrix_to_bytes(D, tmp_path / "e.bin")
    assert np.array_equal(_bytes(tmp_path / "e.bin"), g["dense0"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", IO_CASES)
def test_save_leaves_with_rank_after_device_compression(name, tmp_path):
    """The whole file, ranks included, from the device-compressed H-matrix; the first dense leaf equals the reference's
    matrix_to_bytes dump."""
    p, g = params(name), load(name)
    T, S = build_trees(p)
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], p["sym"], p["uplo"])
    tb.set_low_rank_generator(p["compressor"])
    A = device_generator(p, T, S)
    H = tb.build(A, T, S, p["rank"], p["rank"])
    hm.save_leaves_with_rank(H, tmp_path / "leaves")
    assert np.array_equal(_bytes(tmp_path / "leaves.csv"), g["leaves"])
    g["dense0"].tofile(tmp_path / "d.bin")
    D = hm.bytes_to_matrix(tmp_path / "d.bin")
    first_dense = int(np.nonzero(H.ranks < 0)[0][0])
    assert np.allclose(H.get_block(first_dense), D, rtol=1e-14, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype", [("io_ellipse_n1000_p2", np.float64), ("io_disk2d_n800_symL_p1", np.float64), ("io_ball_n1500_p4_rank2", np.float32)])
def test_operator_binary_dump_roundtrip(name, dtype, tmp_path):
    """HMatrix.save -> HMatrixTreeBuilder.load: same leaves, same blocks bit for bit, same product."""
    p = params(name)
    T, S = build_trees(p)
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], p["sym"], p["uplo"])
    tb.set_low_rank_generator(p["compressor"])
    A = device_generator(p, T, S)
    H = tb.build(A, T, S, p["rank"], p["rank"], dtype=dtype)
    H.save(tmp_path / "op.hmx")
    G = tb.load(tmp_path / "op.hmx", T, S, p["rank"], p["rank"])
    assert G.f32 == H.f32 and np.array_equal(G.ranks, H.ranks)
    for leaf in list(range(0, len(H.ranks), max(1, len(H.ranks) // 25))):
        a, b = H.get_block(leaf), G.get_block(leaf)
        if H.ranks[leaf] >= 0:
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        else:
            assert np.array_equal(a, b)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(H.source_size).astype(dtype)
    y1, y2 = np.zeros(H.target_size, dtype=dtype), np.zeros(H.target_size, dtype=dtype)
    hm.internal_add_hmatrix_vector_product("N", 1.0, H, x, 0.0, y1)
    hm.internal_add_hmatrix_vector_product("N", 1.0, G, x, 0.0, y2)
    assert np.linalg.norm(y1 - y2) <= (1e-5 if dtype == np.float32 else 1e-13) * np.linalg.norm(y1)
    # a file for another block tree is refused
    tb2 = hm.HMatrixTreeBuilder(p["eps"], p["eta"] / 2, p["sym"], p["uplo"])
    with pytest.raises(hm.HmxError, match="different block tree|does not match"):
        tb2.load(tmp_path / "op.hmx", T, S, p["rank"], p["rank"])


def _information_lines(text):
    """The reference's print_tree_parameters + print_hmatrix_information output without the wall-clock / thread-count lines."""
    return [ln for ln in text.splitlines() if not ln.startswith(("Block_tree_walltime", "Blocks_computation_walltime", "Number_of_threads"))]


@pytest.mark.gpu
@pytest.mark.parametrize("name", IO_CASES)
def test_tree_parameters_and_hmatrix_information(name):
    """print_tree_parameters + print_hmatrix_information (what use_hmatrix.cpp prints): byte-identical to the text the
    reference wrote for the same operator, apart from its timing and OpenMP lines."""
    p, g = params(name), load(name)
    T, S = build_trees(p)
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], p["sym"], p["uplo"])
    tb.set_low_rank_generator(p["compressor"])
    H = tb.build(device_generator(p, T, S), T, S, p["rank"], p["rank"])
    out = io.StringIO()
    hm.print_tree_parameters(H, out)
    hm.print_hmatrix_information(H, out)
    assert _information_lines(out.getvalue()) == _information_lines(g["information"].tobytes().decode())
    info = hm.get_hmatrix_information(H)
    assert info["Blocks_computation_walltime"].endswith(" second(s)") and float(info["Compression_ratio"]) > 1
