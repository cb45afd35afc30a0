"""bench.py's multi-rank control flow on ONE GPU: two ranks under torch.distributed.run exactly as the driver launches them, but
sharing device 0 and talking through gloo (RCCL refuses two ranks on one device; HMX_BENCH_SAME_DEVICE / HMX_BENCH_BACKEND are
test hooks).  Checks the row-partitioned build per rank, the graphed local product + collective, the max-over-ranks timing and
that exactly one JSON line with the contract's fields comes out of rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--mu", "4"]])
def test_bench_two_ranks_one_gpu(extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HMX_BENCH_SAME_DEVICE="1", HMX_BENCH_BACKEND="gloo", HMX_BENCH_N="200000")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["unit"] == "GB/s" and d["value"] > 0
    assert "row-partition x2" in d["config"]["parallelism"]


@pytest.mark.parametrize("impl", ["native", "python"])
def test_bench_starts_its_own_ranks(impl):
    """`python bench.py --gpus 2` WITHOUT a launcher: bench.py starts the two ranks itself (fresh children, before anything touches the
    GPU), rank 0's JSON line comes out of the parent.  native: the step is one hmx_dist_* call per product (here over host-staged gloo
    collectives, since the two ranks share the box's one GPU), the exchange variants (0 / 2 / 4 row chunks on the side stream, all-gather /
    broadcasts or pairwise send / recv) are tried and reported; python: the torch.distributed layer."""
    env = dict(os.environ, HMX_BENCH_SAME_DEVICE="1", HMX_BENCH_BACKEND="gloo", HMX_BENCH_N="200000")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dist-impl", impl],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0
    assert len(d["dist"]["per_rank_local_ms"]) == 2 and len(d["dist"]["per_rank_GB"]) == 2
    if impl == "native":
        assert d["dist"]["impl"].startswith("native"), d["dist"]
        # exchange variants: 0 / 2 / 4 row chunks, each as all-gather / broadcasts and pairwise (send / recv)
        assert set(d["dist"]["exchange_trials_ms"]) >= {"0", "2", "0+p2p", "2+p2p"}, d["dist"]
        # ... and the north star's wording of the exchange, ncclAllReduce of the zero-padded output vector, as one more variant
        assert "allreduce" in d["dist"]["exchange_trials_ms"], d["dist"]
        # the number of ranks comes from the communicator, the exposed exchange time from HIP events around the non-local part
        assert d["dist"]["rccl_ranks"] == 2
        assert d["dist"]["exposed_exchange_ms"] is not None and d["dist"]["exposed_exchange_ms"] >= 0 and "HIP events" in d["dist"]["exposed_exchange_method"]
        assert d["dist"]["local_ms_events"] > 0
        # next to the multi-rank number: htool's own MPI + OpenMP path on this box's host cores, same configuration, cores stated
        if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "dist_bench")) and os.path.exists("/opt/conda/bin/mpiexec"):
            cb = d["cpu_baseline"]
            assert cb["kind"] == "reference-mpi" and cb["mpi_ranks"] == 2 and cb["cores"] == 2 * cb["omp_threads_per_rank"] and cb["value"] > 0, cb
            assert cb["n"] == 200000, cb
    else:
        assert d["dist"]["impl"].startswith("python")


def test_bench_eight_ranks_one_gpu_against_the_reference_fixture(tmp_path):
    """The shape of the driver's 8-GPU run -- `bench.py --gpus 8`, eight ranks, the 8-way row partition, hmx_dist_* with every exchange variant
    tried under the watchdog -- on the box's ONE GPU (ranks share device 0, host-staged gloo collectives).  N = 1e5; the product of the
    oracle's hashed vector is compared, through the user numbering, with what htool itself computed for BASELINE configs[1]
    (tests/golden/full_ellipse_n100000.npz: another partition, hence another block tree: the two operators approximate the same matrix to
    epsilon = 1e-4, not to rounding), and the line must explain itself: ranks of the communicator, chosen variant, per-variant times."""
    import numpy as np

    import htool_amd as hm
    from helpers import MANIFEST, load
    env = dict(os.environ, HMX_BENCH_SAME_DEVICE="1", HMX_BENCH_BACKEND="gloo", HMX_BENCH_N="100000", HMX_BENCH_STAGE_TIMEOUT="300")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    dump = str(tmp_path / "y8.npz")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--no-reference", "--dump-product", dump],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["value"] > 0 and "row-partition x8" in d["config"]["parallelism"]
    cfg = d["config"]
    assert cfg["rccl_ranks"] == 8 and cfg["exchange_variant"] in cfg["exchange_trials_ms"] and cfg["watchdog"] is None, cfg
    assert "0" in cfg["exchange_trials_ms"] and d["dist"]["plain_exchange_ms"] > 0
    assert len(d["dist"]["per_rank_local_ms"]) == 8
    # the 8-rank product against the reference's own: same points, same kernel, x = hashed_vector(n, 1) in each tree's cluster numbering
    name = "full_ellipse_n100000"
    p, g = MANIFEST[name], load(name)
    n = p["n"]
    x = hm.create_geometry(p["geom"], n)
    b = hm.ClusterTreeBuilder()
    b.set_maximal_leaf_size(p["leaf"])
    perm_ref = np.asarray(b.create_cluster_tree(n, 3, x, 2, p.get("partitions", 2)).get_permutation())  # the fixture's tree (sha-checked in test_gpu_full_size_reference)
    z = np.load(dump)
    assert int(z["n_gpus"]) == 8 and len(z["y"]) == n
    # bench.py multiplied h = hashed_vector(n, 1) in the cluster numbering of ITS tree (8 partitions: another permutation than the fixture's
    # 2-partition tree).  Two checks: (1) the eight ranks' product equals, to rounding, the product of the same 8-way operator built whole on
    # one GPU; (2) that operator, in USER numbering, applied to the fixture's input reproduces the reference's rows to epsilon.
    from oracle.oracle import hashed_vector
    import bench
    h = hashed_vector(n, 1)
    perm8 = z["perm"]
    x_user = np.empty(n)
    x_user[perm_ref] = h  # the fixture's input in user numbering
    T8 = b.create_cluster_tree(n, 3, x, 2, 8)
    assert np.array_equal(np.asarray(T8.get_permutation()), perm8)
    tb = hm.HMatrixTreeBuilder(p["eps"], p["eta"], "N", "N")
    tb.set_low_rank_generator("partialACA")
    tb.set_minimal_target_depth(bench.minimal_depth(n))  # as bench.py builds it
    tb.set_minimal_source_depth(bench.minimal_depth(n))
    H8 = tb.build(hm.InvDistGenerator(3, x, x, 1e-5, 1.0), T8, T8)
    y8 = np.zeros(n)
    hm.internal_add_hmatrix_vector_product("N", 1.0, H8, h, 0.0, y8)
    assert np.linalg.norm(z["y"] - y8) <= 1e-12 * np.linalg.norm(y8)  # eight row-partitioned ranks + exchange = the whole operator
    yu = np.zeros(n)
    hm.add_hmatrix_vector_product("N", 1.0, H8, x_user, 0.0, yu)  # user numbering in and out
    ref_rows_user = perm_ref[g["rows"]]
    err = np.linalg.norm(yu[ref_rows_user] - g["yN_a1b0"]) / np.linalg.norm(g["yN_a1b0"])
    assert err < 5 * p["eps"], err  # two compressions of the same matrix at epsilon = 1e-4
