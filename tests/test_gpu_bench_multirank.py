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
