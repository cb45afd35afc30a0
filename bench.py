#!/usr/bin/env python3
"""bench.py -- H-matvec throughput of the HIP engine on MI355X (BASELINE.json metric).

A "step" is one H-matrix-vector product y = H x of the whole operator (trans='N', alpha=1, beta=0), input
vector already resident in HBM in cluster ("partition") numbering:
  N = 1 : config "N=1e6 fp64, eta=10, eps=1e-4, 1xMI355X" (BASELINE.json configs[2]; the metric is quoted at
          N=1e6 and it fits one GPU), planar 4:1 ellipse, kernel 1/(1e-5+|x-y|), leaf 100, partialACA,
          minimal block depth 5.
  N > 1 : the same operator row-partitioned over N GPUs (DistributedOperator, configs[3]): local product +
          all-gather of the output slices over RCCL; strong scaling (total work fixed).
value = algorithmic bytes of the whole job / step time, B_alg = 8 * [C_gen + (N_source + N_target)] per rank
(C_gen = htool's number_of_generated_coefficient), summed over ranks, max time over ranks.

Launch: `python bench.py --gpus N` under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment, as the driver
does), or bare -- then bench.py starts its N ranks itself (fresh child processes, before anything touches the GPU).  --gpus must
equal WORLD_SIZE: a mismatch is an error, never a silent 1-GPU run.
N > 1 runs the step through libhmx's C-level DistributedOperator (hmx_dist_*: one C call per product, RCCL communicator of its
own, optional overlap of the output exchange with the expand stage on a side stream -- both variants are timed in the warm-up
phase and reported, the faster one is the timed region); --dist-impl python selects the torch.distributed layer instead.

One JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     : dominant kernel (expand_kernel) algorithmic bytes per launch / its average duration measured
                 with HIP events on the launch stream, against the 8 TB/s HBM3E peak.
  cpu_baseline : the CPU restatement of htool's OpenMP leaf loop (oracle/, "port") timed on this box's host
                 cores on a bounded sample (the block rows of the first 1/16 of the rows) of the SAME
                 compressed operator.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=int(os.environ.get("HMX_BENCH_N", 1000000)))
    ap.add_argument("--geom", default="ellipse")
    ap.add_argument("--eps", type=float, default=1e-4)
    ap.add_argument("--eta", type=float, default=10.0)
    ap.add_argument("--leaf", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--recompress", action="store_true", help="SVD recompression of the ACA output before timing (recompression(hmatrix))")
    ap.add_argument("--sym", default="N", help="symmetry of the builder: N, S (lower storage, sympartialACA), or H (Hermitian, complex dtypes)")
    ap.add_argument("--trans", default="N")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32", "z64", "c32"], help="coefficient type (f32 = htool's HMatrix<float,double>; z64 / c32 = complex double / float, kernel (1+i)/(1e-5+r))")
    ap.add_argument("--force-dist", action="store_true", help="run the row-partition + collective code path even with one rank (testing)")
    ap.add_argument("--emulate-world", type=int, default=0, help="single process: build and time only the block rows of --emulate-rank out of this many partitions (per-rank cost of a multi-GPU run, no collective)")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--mu", type=int, default=1, help="number of right-hand sides (row-major multi-RHS product when > 1)")
    ap.add_argument("--cpu-sample-frac", type=float, default=1.0 / 16)
    ap.add_argument("--dist-impl", default=os.environ.get("HMX_BENCH_DIST_IMPL", "native"), choices=["native", "python"],
                    help="N > 1: hmx_dist_* through one C call per step (native) or the torch.distributed layer (python)")
    ap.add_argument("--generator", default="device", choices=["device", "callback"],
                    help="device: the built-in kernel evaluated on the GPU; callback: the same function as a USER's VirtualGenerator in compiled host code "
                         "(examples/host_generator.c through hmx_hmatrix_set_callback) -- the literal drop-in route of examples/use_hmatrix.cpp")
    ap.add_argument("--callback-threads", type=int, default=0, help="host threads that call the generator (0: all cores, at most 64)")
    ap.add_argument("--no-callback-build", action="store_true", help="do not time a second build of the operator through the host-generator route")
    ap.add_argument("--no-reference", action="store_true", help="skip the timing of htool itself (oracle/_ref/ref_driver) on the host cores")
    ap.add_argument("--output-placement", default="operator", choices=["operator", "torch"], help="where the product's output vector comes from: the operator's own allocator (placed where its sweeps write fastest) or torch.zeros")
    ap.add_argument("--option", action="append", default=[], help="engine option name=value for the operator (hmx_hmatrix_set_option; htool_amd._lib.OPTIONS), e.g. sym_multi_rhs=1")
    ap.add_argument("--dump-product", default=None, help="after the timed region: y = A x for x = the oracle's hashed_vector(n, 1), through the step's own path; rank 0 writes "
                    "y (partition numbering) and the cluster permutation to this .npz (tests compare it with the reference's fixtures)")
    return ap.parse_args()


def spawn_ranks_if_needed(args):
    """`--gpus N` without a launcher: start the N ranks here, as fresh CHILD processes (subprocess.Popen, never os.exec*), BEFORE anything
    initialises the GPU in this process: this function is the first thing main() does, before torch, HIP or libhmx are even imported.  On
    this pool a process that has touched the GPU and then replaces itself with another program takes the machine down; keep it that way --
    nothing below main()'s first line may move above it, and the parent only supervises and relays rank 0's JSON line.  With a launcher,
    --gpus must match WORLD_SIZE."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            print("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d "
                  "bench.py --gpus %d ...) or drop the launcher and let bench.py start its ranks" % (args.gpus, ws, args.gpus, args.gpus), file=sys.stderr)
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs, out0 = [], None
    import tempfile
    # rank 0 writes the line of the first COMPLETE measurement (the plain exchange) here as soon as it exists: whatever happens afterwards
    # -- a rank that aborts inside an untested exchange variant, a hang the watchdog cannot report -- the parent still has a line to relay
    side_fd, side_path = tempfile.mkstemp(prefix="hmx_bench_plain_", suffix=".json")
    os.close(side_fd)

    def complete_line(raw):
        for ln in reversed(raw.decode(errors="replace").splitlines()):
            ln = ln.strip()
            if ln.startswith("{") and ln.endswith("}"):
                try:
                    json.loads(ln)
                    return ln
                except ValueError:
                    pass
        return None

    with tempfile.TemporaryFile() as cap:  # rank 0's stdout (the JSON line); a pipe nobody drains while we poll could fill up
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HMX_BENCH_SIDE_FILE=side_path)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.setdefault("NCCL_SOCKET_IFNAME", "lo")
            procs.append(subprocess.Popen([sys.executable, os.environ.get("HMX_BENCH_CHILD_SCRIPT", os.path.abspath(__file__))] + sys.argv[1:], env=env,  # (the variable: tests of this supervisor with stand-in ranks)
                                          stdout=cap if r == 0 else subprocess.DEVNULL))
        # supervise: the first rank that fails (OOM, RCCL initialisation) takes the others down instead of leaving them in a collective
        # for ever; an overall limit bounds a hang (HMX_BENCH_SPAWN_TIMEOUT seconds, default 3600)
        deadline = time.time() + float(os.environ.get("HMX_BENCH_SPAWN_TIMEOUT", 3600))
        failed = None
        while True:
            rcs = [p.poll() for p in procs]
            if all(rc is not None for rc in rcs):
                break
            bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad or time.time() > deadline:
                failed = ("rank %d exited with code %s" % (bad[0], rcs[bad[0]])) if bad else "time limit reached"
                for p in procs[1:]:
                    if p.poll() is None:
                        p.terminate()
                # rank 0 gets a few seconds for its own way out (its watchdog reports the plain exchange, or it finishes) before it is stopped too
                t_grace = time.time() + float(os.environ.get("HMX_BENCH_RANK0_GRACE", 8))
                while procs[0].poll() is None and time.time() < t_grace:
                    time.sleep(0.1)
                if procs[0].poll() is None:
                    procs[0].terminate()
                t_kill = time.time() + 10
                while any(p.poll() is None for p in procs) and time.time() < t_kill:
                    time.sleep(0.1)
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                break
            time.sleep(0.2)
        rcs = [p.wait() for p in procs]
        cap.seek(0)
        out0 = cap.read()
    try:
        with open(side_path, "rb") as fh:
            side = fh.read()
        os.unlink(side_path)
    except OSError:
        side = b""
    if failed:
        # the measurement that was complete before the failure is not thrown away: rank 0's own line if it got one out, else the side file
        line = complete_line(out0) or complete_line(side)
        print("bench.py: %s; remaining ranks stopped%s" % (failed, "; relaying the measurement completed before the failure" if line else ""), file=sys.stderr)
        if line:
            sys.stdout.write(line + "\n")
            sys.stdout.flush()
        sys.exit(1)
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    sys.exit(max(abs(rc) for rc in rcs))


def kernel_sources_hash():
    """sha256 over the device sources: profiles/traffic.json records it, a PMC measurement of an older binary is not reported."""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels_common.hpp", "kernels_body.hpp", "kernels_compress.hpp", "kernels_pack.hpp", "kernels_matvec.hpp", "kernels_multi_rhs.hpp", "kernels_symmetric.hpp", "kernels_util.hpp", "engine_body.hpp", "engine_state.hpp", "engine_layout.hpp", "engine_products.hpp", "engine_build.hpp", "engine_access.hpp", "engine_entry.hpp", "engine_common.hpp", "engine_api.hpp", "engine_inst.hip", "engine.hip"):
        with open(os.path.join(ROOT, "htool_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def host_generator(hm, x, np_dt, cplx, herm, threads):
    """examples/host_generator.c (examples/libhostgen.so, built by __graft_entry__.build()) as an htool_amd.NativeGenerator"""
    import ctypes as C

    class InvDist(C.Structure):
        _fields_ = [("dim", C.c_int32), ("pad", C.c_int32), ("target", C.c_void_p), ("source", C.c_void_p), ("delta", C.c_double), ("scale", C.c_double),
                    ("cre", C.c_double), ("cim", C.c_double), ("hermitian", C.c_int32), ("pad2", C.c_int32)]
    lib = C.CDLL(os.path.join(ROOT, "examples", "libhostgen.so"))
    xc = np.ascontiguousarray(x, dtype=np.float64)
    g = InvDist(3, 0, xc.ctypes.data, xc.ctypes.data, 1e-5, 1.0, 1.0, 1.0 if cplx else 0.0, int(bool(herm)), 0)
    fn = {np.dtype(np.float64): lib.hostgen_inv_dist_f64, np.dtype(np.float32): lib.hostgen_inv_dist_f32,
          np.dtype(np.complex128): lib.hostgen_inv_dist_z64, np.dtype(np.complex64): lib.hostgen_inv_dist_c32}[np.dtype(np_dt)]
    return hm.NativeGenerator(fn, C.addressof(g), threads=threads, keep=(lib, g, xc))


def minimal_depth(n):
    """Smallest d with n / 2^d <= 46340 (no admissible block can reach M*N >= 2^31)."""
    d = 0
    while n / (2 ** d) > 46340:
        d += 1
    return d


def cpu_baseline(H, T, frac, log):
    """Time the CPU restatement of openmp_internal_add_hmatrix_vector_product on a row slab of the operator."""
    from oracle import oracle as O
    tab = H.leaf_table()
    n = T.get_size()
    nodes = T.nodes_int()
    # row slab [0, cut): the smallest cluster at offset 0 that is at least as large as the largest leaf's
    # target cluster and as frac*n -- no leaf straddles it, so rows [0,cut) depend on the selected leaves only
    need = max(int(tab[:, 1].max()), int(frac * n))
    cand = nodes[(nodes[:, 1] == 0) & (nodes[:, 2] >= need)]
    cut = int(cand[:, 2].min())
    sel = np.nonzero(tab[:, 0] + tab[:, 1] <= cut)[0]
    t0 = time.time()
    data, offs, pos = [], [], 0
    for b in sel:
        blk = H.get_block(int(b))
        if tab[b, 4] >= 0:
            u, v = np.asfortranarray(blk[0]).ravel("F"), np.asfortranarray(blk[1]).ravel("F")
            offs.append((pos, pos + u.size))
            data += [u, v]
            pos += u.size + v.size
        else:
            d = np.asfortranarray(blk).ravel("F")
            offs.append((pos, 0))
            data.append(d)
            pos += d.size
    log("cpu_baseline: downloaded %d leaves (%.2f GB) in %.1fs" % (len(sel), pos * 8 / 1e9, time.time() - t0))
    Ho = O.HMatrix.from_blocks(tab[sel], np.array(offs), np.concatenate(data), [0, cut, 0, H.source_size])
    # the cores this process really has (cgroup CPU quota), as for the reference: 128 OpenMP threads on a 16-core quota oversubscribe 8 x
    import htool_amd as _hm
    if hasattr(O.lib(), "orc_set_num_threads"):
        O.lib().orc_set_num_threads(int(_hm.lib().hmx_host_cores()))
    x = O.hashed_vector(H.source_size, 1)
    best = 1e30
    y = None
    for _ in range(4):
        t = time.time()
        y = Ho.matvec(x, "N", 1.0, 0.0, policy="omp")
        best = min(best, time.time() - t)
    bytes_alg = 8.0 * (pos + H.source_size + cut)
    return dict(value=bytes_alg / best / 1e9, unit="GB/s", cores=int(O.lib().orc_num_threads()), kind="port",
                sample="block rows [0,%d) of the same compressed operator (%d leaves, %.2f GB), "
                       "openmp leaf loop, best of 4" % (cut, len(sel), pos * 8 / 1e9)), y, cut


def reference_baseline(log, n, geom, eps, eta, leaf, depth, cores=None):
    """htool ITSELF (oracle/_ref/ref_driver: the real headers + the image's MKL, OpenMP policy: openmp_build +
    openmp_internal_add_hmatrix_vector_product) timed on this box's host cores on THE SAME configuration as the GPU run (geometry,
    N, kernel, eps, eta, leaf size, minimal block depth).  Time-boxed (HMX_BENCH_REF_TIMEOUT seconds, default 240: the N=1e6 build
    takes 150 s on 8 cores, ~20 s on 64); when the box is too slow for that the N=1e5 operator (configs[1]) is timed instead and the
    sample says so.  Only runs where the binary built in the dev container travelled with the repo; never touches /root/reference."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not os.path.exists(exe):
        return None
    budget = float(os.environ.get("HMX_BENCH_REF_TIMEOUT", 240))
    ncpu = cores or os.cpu_count() or 1  # the cores the process really has (cgroup quota: hmx_host_cores), not the hardware threads

    def run(nn, dd, threads, reps, timeout):
        t0 = time.time()
        out = subprocess.run([exe, "hmat", "n=%d" % nn, "geom=%s" % geom, "eps=%g" % eps, "eta=%g" % eta, "leaf=%d" % leaf, "mindepth=%d" % dd,
                              "compressor=partialACA", "par=1", "time_reps=%d" % reps, "dump_blocks=0", "out=/dev/null"], capture_output=True, text=True,
                             timeout=timeout, env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
        m = re.search(r"cgen=(\d+)\+(\d+).*tree=([0-9.]+)s build=([0-9.]+)s matvec=([0-9.]+)s", out.stdout)
        if not m:
            log("reference driver produced no timing: %s" % (out.stdout[-200:] + out.stderr[-200:]))
            return None
        log("reference (htool, OpenMP, %d threads) N=%d: tree %.1fs build %.2fs matvec %.4fs (%.1fs wall)" % (threads, nn, float(m.group(3)), float(m.group(4)), float(m.group(5)), time.time() - t0))
        return dict(cgen=int(m.group(1)) + int(m.group(2)), build_s=float(m.group(4)), matvec_s=float(m.group(5)), threads=threads)

    try:
        best, size = None, n
        t_start = time.time()
        # the reference's per-thread temporaries make "all cores" a poor choice for the product: 64 threads first, then 16 if time is left
        for threads in sorted({min(64, ncpu), min(16, ncpu)}, reverse=True):
            left = budget - (time.time() - t_start)
            if left < (30 if best else 5):
                break
            try:
                r = run(n, depth, threads, 5, left)
            except subprocess.TimeoutExpired:
                log("reference at N=%d did not finish in the time box (%d threads)" % (n, threads))
                break
            if r and (best is None or r["matvec_s"] < best["matvec_s"]):
                best = r
        if best is None and n > 100000:  # fall back to configs[1]
            size = 100000
            for threads in sorted({min(64, ncpu), min(16, ncpu)}, reverse=True):
                r = run(size, 0, threads, 10, 120)
                if r and (best is None or r["matvec_s"] < best["matvec_s"]):
                    best = r
        if best is None:
            return None
        return dict(value=8.0 * (best["cgen"] + 2.0 * size) / best["matvec_s"] / 1e9, unit="GB/s", cores=best["threads"], kind="reference",
                    sample="htool itself (openmp_internal_add_hmatrix_vector_product, MKL sequential BLAS) on the %s: N=%d %s, eps=%g, eta=%g, leaf %d, "
                           "min block depth %d; best of 5 products" % ("same configuration" if size == n else "configs[1] operator (the N=%d build did not fit the time box)" % n,
                                                                        size, geom, eps, eta, leaf, depth if size == n else 0),
                    build_s=best["build_s"], matvec_s=best["matvec_s"], n=size, comparable=bool(size == n))
    except Exception as e:
        log("reference driver failed: %r" % (e,))
        return None


def reference_mpi_baseline(log, world, n, geom, eps, eta, leaf, depth, cores=None):
    """htool's own MPI + OpenMP CPU path next to the multi-GPU numbers: oracle/_ref/dist_bench (the real headers + MPICH + MKL,
    built in the dev container) under `mpiexec -n world`, cores / world OpenMP threads per rank, on THE SAME configuration --
    every rank builds its block rows (openmp_build), the product is internal_add_distributed_operator_vector_product_global_to_global
    (openmp leaf loop per rank + MPI_Allgatherv), best of 5, maximum over the ranks.  Time-boxed (HMX_BENCH_REF_TIMEOUT, default
    240 s; falls back to the N=1e5 operator).  Started as a child process by rank 0 AFTER its process group is gone."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "dist_bench")
    mpiexec = os.environ.get("HMX_BENCH_MPIEXEC", "/opt/conda/bin/mpiexec")
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        log("no oracle/_ref/dist_bench or mpiexec here: no reference-mpi baseline")
        return None
    budget = float(os.environ.get("HMX_BENCH_REF_TIMEOUT", 240))
    threads = max(1, (cores or os.cpu_count() or world) // world)  # cores: what the process really has (cgroup quota)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "libs") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)

    def run(nn, dd, timeout):
        t0 = time.time()
        out = subprocess.run([mpiexec, "-n", str(world), exe, "n=%d" % nn, "geom=%s" % geom, "eps=%g" % eps, "eta=%g" % eta, "leaf=%d" % leaf,
                              "mindepth=%d" % dd, "compressor=partialACA", "reps=5"], capture_output=True, text=True, timeout=timeout, env=env)
        m = re.search(r"ranks=(\d+) threads=(\d+) n=\d+ cgen=(\d+) build=([0-9.]+)s matvec=([0-9.]+)s", out.stdout)
        if not m:
            log("reference-mpi produced no timing: %s" % (out.stdout[-200:] + out.stderr[-300:]))
            return None
        log("reference (htool, MPI x OpenMP = %s x %s) N=%d: build %.2fs, product %.5fs (%.1fs wall)" % (m.group(1), m.group(2), nn, float(m.group(4)), float(m.group(5)), time.time() - t0))
        return dict(ranks=int(m.group(1)), threads=int(m.group(2)), cgen=int(m.group(3)), build_s=float(m.group(4)), matvec_s=float(m.group(5)))

    try:
        size, r = n, None
        try:
            r = run(n, depth, budget)
        except subprocess.TimeoutExpired:
            log("reference-mpi at N=%d did not finish in the time box" % n)
        if r is None and n > 100000:
            size = 100000
            r = run(size, 0, 120)
        if r is None:
            return None
        return dict(value=8.0 * (r["cgen"] + world * size + size) / r["matvec_s"] / 1e9, unit="GB/s", cores=r["ranks"] * r["threads"], kind="reference-mpi",
                    sample="htool itself: DistributedOperator over %d MPI ranks x %d OpenMP threads (internal_add_distributed_operator_vector_product_global_to_global, "
                           "MKL sequential BLAS, MPICH) on the %s: N=%d %s, eps=%g, eta=%g, leaf %d, min block depth %d; best of 5 products, max over ranks"
                           % (r["ranks"], r["threads"], "same configuration" if size == n else "configs[1] operator (the N=%d run did not fit the time box)" % n,
                              size, geom, eps, eta, leaf, depth if size == n else 0),
                    build_s=r["build_s"], matvec_s=r["matvec_s"], n=size, mpi_ranks=r["ranks"], omp_threads_per_rank=r["threads"])
    except Exception as e:  # noqa: BLE001
        log("reference-mpi failed: %r" % (e,))
        return None


class Watchdog:
    """First contact with a real multi-GPU node must not cost the measurement.  The exchange variants of the distributed product (row chunks
    on a side stream, pairwise send / recv, all-reduce) have only ever run on one GPU; a collective that hangs cannot be cancelled from
    inside the process.  Every rank arms this before a stage that contains collectives; when the stage does not finish in time the rank that
    holds `emit` (rank 0) writes the JSON line of the last COMPLETE measurement -- the plain exchange, timed before any variant is tried --
    and every rank leaves with os._exit.  Without a complete measurement the exit code says so."""

    def __init__(self, log):
        import threading
        self.log, self.lock = log, threading.Lock()
        self.deadline, self.stage, self.fallback = None, None, None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def arm(self, stage, seconds):
        with self.lock:
            self.stage, self.deadline = stage, time.time() + seconds

    def disarm(self):
        with self.lock:
            self.deadline = None

    def set_fallback(self, fn):
        with self.lock:
            self.fallback = fn

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                late = self.deadline is not None and time.time() > self.deadline
                stage, fb = self.stage, self.fallback
            if late:
                self.log("WATCHDOG: stage '%s' did not finish in time -- %s" % (stage, "reporting the plain exchange measured before it" if fb else "no complete measurement yet"))
                # the fallback is host-only code (it formats numbers measured earlier); should it block all the same, this second timer ends
                # the process: the parent then relays the side file the plain measurement was written to (spawn_ranks_if_needed)
                import threading
                threading.Timer(5.0, lambda: os._exit(4)).start()
                try:
                    if fb:
                        fb(stage)
                finally:
                    os._exit(0 if fb else 4)


def main():
    args = parse()
    spawn_ranks_if_needed(args)
    # RCCL / HIP runtime banners go to the C-level stdout; keep fd 1 clean for the single JSON line
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # the host driver of this platform only supports dmabuf IPC: RCCL / device-memory sharing between the ranks' processes needs it
    # (already exported on the GPU boxes; kept here so that a bare environment works too).  Must be set before HIP initialises.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    import torch
    import torch.distributed as dist
    import htool_amd as hm
    from htool_amd import distributed as D

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    if os.environ.get("HMX_BENCH_SAME_DEVICE"):  # test hook: several ranks on one GPU (with HMX_BENCH_BACKEND=gloo; RCCL refuses that)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        backend = os.environ.get("HMX_BENCH_BACKEND", "nccl")
        # one node by contract: RCCL's bootstrap sockets stay on the loopback interface (no dependence on whatever other
        # interfaces the box has; the data path is xGMI / shared memory either way)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n = args.n
    t0 = time.time()
    x = hm.create_geometry(args.geom, n)  # the synthetic input itself: not part of the tree build
    t_geom = time.time() - t0
    t0 = time.time()
    ctb = hm.ClusterTreeBuilder()
    ctb.set_maximal_leaf_size(args.leaf)
    # single GPU: HMatrixBuilder's default of 2 partitions (hmatrix/utility.hpp:23), whole operator on the GPU
    emu = args.emulate_world
    T = ctb.create_cluster_tree(n, 3, x, 2, emu if emu else (world if use_dist else 2))
    t_tree = time.time() - t0
    tb = hm.HMatrixTreeBuilder(args.eps, args.eta, args.sym, "L" if args.sym != "N" else "N")
    tb.set_low_rank_generator("partialACA" if args.sym == "N" else "sympartialACA")
    d = minimal_depth(n)
    tb.set_minimal_target_depth(d)
    tb.set_minimal_source_depth(d)
    for kv in args.option:
        tb.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    cplx = args.dtype in ("z64", "c32")
    t_init = time.time()
    hm.lib().hmx_device_init(local_rank)  # HIP context + load of libhmx's code object: not part of an operator build
    t_init = time.time() - t_init
    # One slab from the driver before anything is timed (hmx_device_reserve): hipMalloc stalls for seconds while the driver scrubs
    # what the PREVIOUS process released (tools/malloc_after_exit.hip), and a build allocates its two largest arrays right there.
    # 60 % of what is free (round 5; before: 64 KB per point): the slab is also where the library looks for a place for the small arrays its
    # sweeps WRITE -- a write stream costs a streaming read 16-23 % when both lie in the same third of the physical memory, 7-10 % otherwise
    # (hmx_option place_written, tools/placement_rw.hip), and a slab of a third of the memory or less is often all of one kind;
    # whatever does not fit is allocated as before, `compress.malloc_s` says what hipMalloc still cost.  HMX_BENCH_RESERVE_GB=0: off.
    free_b, _ = torch.cuda.mem_get_info(local_rank)
    share = max(1, world if os.environ.get("HMX_BENCH_SAME_DEVICE") else 1)
    want = float(os.environ["HMX_BENCH_RESERVE_GB"]) * 1e9 if "HMX_BENCH_RESERVE_GB" in os.environ else 0.6 * free_b / share
    reserve_b = int(min(0.6 * free_b / share, want))
    t_res = time.time()
    reserved = reserve_b >= (1 << 30) and hm.lib().hmx_device_reserve(local_rank, reserve_b) == 0
    t_res = time.time() - t_res
    gen = hm.InvDistGenerator(3, x, x, 1e-5, 1.0, 1.0, 1.0 if cplx else 0.0, args.sym == "H")
    t0 = time.time()
    part = use_dist
    brank = args.emulate_rank if emu else (rank if part else -1)
    np_dt = {"f64": np.float64, "f32": np.float32, "z64": np.complex128, "c32": np.complex64}[args.dtype]
    t_dt = {"f64": torch.float64, "f32": torch.float32, "z64": torch.complex128, "c32": torch.complex64}[args.dtype]
    esz = float(np.dtype(np_dt).itemsize)
    malloc0 = hm.lib().hmx_device_malloc_seconds()
    if args.generator == "callback":
        gen = host_generator(hm, x, np_dt, cplx, args.sym == "H", args.callback_threads)
    H = tb.build(gen, T, T, brank, brank, device=local_rank, dtype=np_dt)
    if args.recompress:
        tr = time.time()
        before = H.stats()
        H.recompress()
        torch.cuda.synchronize()
        log("recompression: %.2fs, mean rank %.2f -> %.2f, C_gen low-rank %.3e -> %.3e" % (time.time() - tr, before["rank_mean"], H.stats()["rank_mean"],
                                                                                        before["cgen_lowrank"], H.stats()["cgen_lowrank"]))
    torch.cuda.synchronize()
    t_build = time.time() - t0
    t_malloc = hm.lib().hmx_device_malloc_seconds() - malloc0
    st = H.stats()
    log("cluster tree %.2fs, device build %.2fs (ACA %.2fs, pack+assemble %.2fs): %d dense + %d low-rank leaves, rank %d/%.2f/%d, "
        "C_gen %.3e + %.3e, %.2f GB in HBM" % (t_tree, t_build, st["t_compress_s"], st["t_pack_s"], st["n_dense"], st["n_lowrank"],
                                               st["rank_min"], st["rank_mean"], st["rank_max"], st["cgen_dense"], st["cgen_lowrank"],
                                               st["stream_bytes"] / 1e9))

    tp = D.PartitionFromCluster(T)
    A = D.DistributedOperator(tp, tp)
    A.add_global_to_local_operator(D.RestrictedGlobalToLocalHMatrix(H))
    xin = torch.from_numpy(np.random.default_rng(1).random(n).astype(np_dt)).to(dev)  # partition numbering, resident in HBM
    y = torch.zeros(n, dtype=t_dt, device=dev)

    def output_buffer(shape):
        """The product's output: resident in HBM either way; by default from the operator (hmx_hmatrix_alloc_vector: where its sweeps write
        fastest -- the expand kernels' two speeds are where y lies relative to the E-streams), --output-placement torch: torch.zeros"""
        if args.output_placement == "operator":
            try:
                return H.empty_output(shape, args.trans if args.trans in ("N", "T", "C") else "N")
            except Exception as e:  # noqa: BLE001
                log("operator-placed output unavailable (%r): torch.zeros" % (e,))
        return torch.zeros(shape, dtype=t_dt, device=dev)
    y_loc = output_buffer(H.nb_rows())

    mu = args.mu
    if mu > 1:
        Xmu = torch.from_numpy(np.random.default_rng(2).random((n, mu)).astype(np_dt)).to(dev)
        Ymu = output_buffer((H.nb_rows(), mu))

    # N > 1, single vector: the C-level DistributedOperator (one C call per step, its own RCCL communicator).  Every rank must take
    # the same path: the outcome of the set-up is agreed on over the process group before anything is timed.
    native, dist_info = None, dict(impl="python (torch.distributed)")
    if part and args.trans == "N" and args.dist_impl == "native":
        ok = 1
        try:
            comm = D.NativeCommunicator(backend="gloo" if os.environ.get("HMX_BENCH_BACKEND", "nccl") != "nccl" else "rccl")
            native = D.NativeDistributedOperator(H, T, T, comm)
        except Exception as e:  # noqa: BLE001 -- any failure means: use the torch.distributed layer
            print("[bench] rank %d: native distributed operator unavailable (%r)" % (rank, e), file=sys.stderr, flush=True)
            ok = 0
        flag = torch.tensor([ok], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            native = None
        else:
            # the number of ranks as the COMMUNICATOR reports it (ncclCommCount), not what the launcher said
            dist_info = dict(impl="native (hmx_dist_*, one C call per step)", rccl_ranks=comm.count(), communicator=comm.backend)

    if mu > 1 and part:
        Yg = torch.zeros((n, mu), dtype=t_dt, device=dev)  # every rank receives the whole result (global-to-global contract)

    def step():
        if mu > 1 and native is not None:  # BASELINE config 5: row-partitioned multi-RHS product, exchange of the mu-interleaved row slices
            native.matmat_row_major_global_to_global("N", 1.0, Xmu, 0.0, Yg, mu)
        elif mu > 1 and part:
            D.internal_add_distributed_operator_matrix_product_row_major_global_to_global("N", 1.0, A, Xmu, 0.0, Yg, mu)
        elif mu > 1:
            hm.internal_add_hmatrix_matrix_product_row_major(args.trans, 1.0, H, Xmu, 0.0, Ymu, mu)
        elif native is not None:
            native.matvec_global_to_global("N", 1.0, xin, 0.0, y)
        elif part:
            D.internal_add_distributed_operator_vector_product_global_to_global("N", 1.0, A, xin, 0.0, y)
        else:
            hm.internal_add_hmatrix_vector_product(args.trans, 1.0, H, xin, 0.0, y_loc)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # Row-partitioned runs are launch-bound per rank (a few hundred microseconds of kernels at 8 GPUs): the local kernels
    # of a step are captured once in a HIP graph (htool_amd.distributed.GraphedGlobalToGlobalProduct); the RCCL all-gather
    # is issued eagerly after each replay, so no graph ever holds a collective.  HMX_BENCH_NO_GRAPH=1: eager launches.
    # ---- measurement helpers ---------------------------------------------------------------------------------------------------------
    b_alg = torch.tensor([esz * (st["cgen_dense"] + st["cgen_lowrank"] + mu * (n + H.nb_rows()))], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(b_alg, op=dist.ReduceOp.SUM)
    # a HOST number from here on: the watchdog's fallback assembles the JSON line while a collective may hang on the device -- nothing on
    # that path may touch torch or HIP (a device-to-host copy would wait for the stream the hung kernel sits on)
    b_alg_f = float(b_alg.item())
    del b_alg
    wd = Watchdog(log) if use_dist else None
    stage_limit = float(os.environ.get("HMX_BENCH_STAGE_TIMEOUT", 180))  # seconds a stage with collectives may take before the watchdog reports the last complete measurement

    def guarded(stage, fn, seconds=None):
        if wd is not None:
            wd.arm(stage, seconds or stage_limit)
        try:
            return fn()
        finally:
            if wd is not None:
                wd.disarm()

    def measure(nsteps):
        """2 untimed steps, then EXACTLY nsteps between barrier + synchronize on both sides; ms per step, maximum over the ranks."""
        for _ in range(2):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        fence()
        tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item()) / nsteps * 1e3

    b_alg_local = esz * (st["cgen_dense"] + st["cgen_lowrank"] + mu * (n + H.nb_rows()))  # this rank's operator alone

    def local_roofline():
        """Roofline of the dominant kernel of the rank-LOCAL product (HIP events on the launch stream; no collective inside)."""
        # ---- roofline of the dominant kernel: HIP events on the launch stream, same steps ------------------------
        H.set_profiling(True)
        acc = {}
        nprof = max(3, min(args.steps, 10))
        for _ in range(nprof):
            if mu > 1:
                hm.internal_add_hmatrix_matrix_product_row_major(args.trans, 1.0, H, Xmu, 0.0, Ymu, mu)
            else:
                hm.internal_add_hmatrix_vector_product(args.trans, 1.0, H, xin, 0.0, y_loc)
            for name, ms in H.last_kernel_times():
                acc.setdefault(name, []).append(ms)
        H.set_profiling(False)
        kern_ms = {k: float(np.mean(v)) for k, v in acc.items()}
        # per launch: the stream once, and per right-hand side the operand vectors (a, x) and the result
        exp_bytes = esz * (st["expand_coeffs"] + mu * (st["a_total"] + n + H.nb_rows()))
        red_bytes = esz * (st["reduce_coeffs"] + mu * (st["a_total"] + n))
        # single vector: expand_kernel / reduce_kernel; multi-RHS: the *_mu (LDS operand), *_mus (scalar operand) or *_mfma16 variants
        exp_name = next((k for k in kern_ms if k.startswith("expand")), "expand_kernel")
        red_name = next((k for k in kern_ms if k.startswith("reduce")), None) or next((k for k in kern_ms if k.startswith("rowsym")), "reduce_kernel")  # (the sweep over the R-streams: rowsym_kernel in a transposed product on the stored data)
        exp_ms = kern_ms.get(exp_name, float("nan"))
        achieved = exp_bytes / (exp_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: measured separately with rocprofv3 --pmc and stored with the sha256 of the kernel
        # sources it was measured on -- a number collected on other kernels is not reported
        traffic = traffic_source = product_traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        std = world == 1 and n == 1000000 and args.geom == "ellipse" and not emu and not use_dist and args.dtype == "f64" and args.eps == 1e-4
        c5 = (world == 1 and n == 4000000 and args.geom == "ellipse" and emu == 8 and args.emulate_rank == 3 and args.dtype == "f32" and args.sym == "S" and mu == 16
              and args.eps == 1e-6 and args.trans == "N")
        if os.path.exists(tf) and (std or c5):
            rec = json.load(open(tf))
            # (dominant kernel's bytes per launch, the whole product's bytes) of the workloads the PMC passes cover
            keys = {("N", "N", 1): ("expand_kernel_hbm_bytes_per_launch", "product_hbm_bytes_total"), ("S", "N", 1): ("expand_sym_kernel_hbm_bytes_per_launch", "sym_product_hbm_bytes_total"),
                    ("N", "N", 16): ("mu16_expand_kernel_hbm_bytes_per_launch", "mu16_product_hbm_bytes_total"), ("N", "T", 1): ("transT_colsum_kernel_hbm_bytes_per_launch", "transT_product_hbm_bytes_total"),
                    ("S", "N", 16): ("sym_mu16_expand_kernel_hbm_bytes_per_launch", "sym_mu16_product_hbm_bytes_total")}
            kk = ("c5_rank3_expand_kernel_hbm_bytes_per_launch", "c5_rank3_product_hbm_bytes_total") if c5 else keys.get((args.sym, args.trans, mu))
            # the counters were collected on the default path of each workload: a line whose engine options choose another path (--option
            # sym_multi_rhs=0: the expanded view, other kernels) must not carry them
            expected = {"N": {1: "expand_kernel", 16: "expand_mfma16s_kernel"}, "S": {1: "expand_sym_kernel", 16: "expand_sym_mfma16_kernel"}}
            want = "expand_colsum_kernel" if args.trans == "T" else expected.get(args.sym, {}).get(mu)
            if args.option or exp_name != want:
                kk = None
            if rec.get("kernel_sources_sha256") == kernel_sources_hash():
                if kk:
                    traffic, product_traffic = rec.get(kk[0]), rec.get(kk[1])
                    traffic_source = "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE (x 2 on gfx950) + WRITE_SIZE, separate passes, round %s, kernel sources sha256 %s" % (rec.get("round"), rec["kernel_sources_sha256"][:16])
            else:
                log("profiles/traffic.json was measured on other kernel sources (hash differs): roofline.traffic = null")
        roofline = dict(bound="hbm", kernel=exp_name, achieved=achieved, peak=8000.0, unit="GB/s", frac=achieved / 8000.0,
                        traffic=traffic, traffic_source=traffic_source, algorithmic_bytes_per_launch=exp_bytes, avg_launch_ms=exp_ms,
                        kernels_ms=kern_ms, reduce_kernel_GBps=(red_bytes / (kern_ms[red_name] * 1e-3) / 1e9) if kern_ms.get(red_name) else None)
        if product_traffic:  # the whole product: HBM bytes all its kernels moved by the counters over the algorithmic bytes of the line (1: nothing read twice, nothing extra)
            roofline["product_traffic"] = product_traffic
            roofline["moved_over_algorithmic"] = product_traffic / b_alg_local

        # measured device copy bandwidth (16 B/lane copy kernel, read+write) as the practical HBM ceiling on this box
        import ctypes
        bw = ctypes.c_double(0.0)
        hm.lib().hmx_device_copy_bandwidth(local_rank, 2 << 30, 5, ctypes.byref(bw))
        roofline["measured_copy_GBps"] = bw.value  # (read + write: no ceiling for a sweep that reads; the read-only rate below is)
        rbw = ctypes.c_double(0)
        hm.lib().hmx_device_read_bandwidth(local_rank, 8 << 30, 5, ctypes.byref(rbw))  # what a pure streaming read reaches on this box
        roofline["measured_read_GBps"] = rbw.value
        roofline["frac_of_measured_read"] = achieved / rbw.value if rbw.value > 0 else None

        return roofline, kern_ms, nprof

    def compress_info():
        # compression as throughput (entries/s, not roofline): kernel entries the ACA evaluated per second of ACA kernel time,
        # dense entries per second of packing kernels; the rest of the device build is host layout work + allocations
        t_aca, t_packk = st["t_compress_s"], st.get("t_assemble_s", 0.0)
        compress = dict(cross_entries_per_s=(st["cgen_lowrank"] / t_aca) if t_aca > 0 else None,
                        dense_entries_per_s=(st["cgen_dense"] / t_packk) if t_packk > 0 else None,
                        aca_kernels_s=t_aca, pack_kernels_s=t_packk, host_s=max(0.0, t_build - t_aca - t_packk - t_malloc), malloc_s=t_malloc, device_total_s=t_build,
                        reserved_slab_GB=reserve_b / 1e9 if reserved else 0.0, reserve_s=t_res,
                        # nothing left out: cluster tree on the host + device initialisation + the slab reservation (the hipMalloc stall the
                        # timed build no longer pays) + the device build; and the operator's stored coefficients per second of all that
                        device_init_s=t_init, device_total_with_reserve_s=t_build + t_res, end_to_end_s=t_tree + t_init + t_res + t_build,
                        entries_per_s_end_to_end=(st["cgen_dense"] + st["cgen_lowrank"]) / (t_tree + t_init + t_res + t_build),
                        # hmx_option place_written: GB/s of the placement probe for the array the reduce stage writes (0: nothing tried)
                        written_array_placement=dict(stream_alone_GBps=st.get("placed_read_gbps"), first_fit_GBps=st.get("placed_first_gbps"),
                                                     chosen_GBps=st.get("placed_gbps"), places_tried=st.get("placed_tried")))

        return compress

    def assemble(ms_per_step, dist_info, graphed):
        value = b_alg_f / (ms_per_step * 1e-3) / 1e9
        cfg = dict(mu=mu, sym=args.sym, trans=args.trans, recompressed=bool(args.recompress), workload="H-matvec N=%d %s, eta=%g, %s eps=%g, leaf %d, %s, kernel 1/(1e-5+r), min block depth %d" % (n, {"f64": "fp64", "f32": "fp32", "z64": "complex fp64", "c32": "complex fp32"}[args.dtype], args.eta, "partialACA" if args.sym == "N" else "sympartialACA (S,L)", args.eps, args.leaf, args.geom, d),
                   parallelism=("row-partition x%d + all-gather%s" % (world, ", local kernels replayed from a HIP graph" if graphed else (", hmx_dist_* (C)" if native is not None else ""))) if part else "single GPU",
                   n_dense=int(st["n_dense"]), n_lowrank=int(st["n_lowrank"]), rank_mean=st["rank_mean"],
                   algorithmic_GB=b_alg_f / 1e9, hbm_roofline_frac=value / (8000.0 * world),
                   build_s=dict(geometry=t_geom, cluster_tree=t_tree, device_total=t_build, aca=st["t_compress_s"],
                                **{k: round(v, 4) for k, v in getattr(H, "_build_walltimes", {}).items()}))
        if args.option:
            cfg["options"] = list(args.option)
        cfg["output_placement"] = args.output_placement
        if use_dist:  # a SCALE record explains itself: how many ranks the communicator really has, which exchange ran, what each variant cost
            cfg.update(rccl_ranks=dist_info.get("rccl_ranks"), communicator=dist_info.get("communicator"), dist_impl=dist_info.get("impl"),
                       exchange_variant=dist_info.get("exchange_variant", "all-gather after the product"), exchange_trials_ms=dist_info.get("exchange_trials_ms"),
                       exchange_failures=dist_info.get("exchange_failures"), watchdog=dist_info.get("watchdog"))
        o = dict(metric="hmatvec_effective_throughput", value=value, unit="GB/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                 ms_per_step=ms_per_step, higher_is_better=True, scaling="strong", vs_baseline=None, dtype=args.dtype, data="synthetic",
                 config=cfg, roofline=roofline, compress=compress)
        if use_dist:
            o["dist"] = dist_info
        return o

    def json_clean(v):  # strict JSON: no NaN / Infinity
        if isinstance(v, dict):
            return {k: json_clean(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [json_clean(x) for x in v]
        if isinstance(v, float) and (v != v or v in (float("inf"), float("-inf"))):
            return None
        return v

    def emit(o):
        os.write(json_fd, (json.dumps(json_clean(o)) + "\n").encode())


    roofline = compress = None
    if use_dist:  # everything the JSON line needs besides the timing exists BEFORE the first exchange variant is tried (Watchdog)
        roofline, kern_ms, nprof = local_roofline()
        compress = compress_info()
        compress["generator"] = args.generator

    graphed = False
    if native is not None:
        # untimed: the result must equal the torch.distributed layer's, then the variants of the output exchange are tried for a few
        # steps each and the fastest becomes the timed region: one exchange after the product or the expand stage in
        # 2 / 4 row chunks with every chunk's exchange on a side stream under the next chunk's kernel; each as an all-gather / grouped
        # broadcasts or pairwise (grouped ncclSend / ncclRecv: one xGMI link per pair on a fully connected node).
        # HMX_DIST_OVERLAP=<chunks> and HMX_DIST_P2P=0/1 pin the choice.
        flag_dev = dev if backend == "nccl" else "cpu"
        out = Yg if mu > 1 else y
        ref = torch.zeros_like(out)
        def torch_reference():
            if mu > 1:
                D.internal_add_distributed_operator_matrix_product_row_major_global_to_global("N", 1.0, A, Xmu, 0.0, ref, mu)
            else:
                D.internal_add_distributed_operator_vector_product_global_to_global("N", 1.0, A, xin, 0.0, ref)
            torch.cuda.synchronize()
        guarded("reference product of the torch.distributed layer", torch_reference)

        def reproduces():
            out.zero_()
            step()
            torch.cuda.synchronize()
            good = torch.tensor([1 if torch.equal(out, ref) else 0], device=flag_dev)
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
            return int(good.item()) == 1

        if not guarded("first native product", reproduces):
            log("native distributed product differs from the torch.distributed layer: falling back")
            native, dist_info = None, dict(impl="python (torch.distributed); native result mismatch")
        else:
            # FIRST a complete measurement of the plain exchange (one all-gather after the product: the reference's MPI_Allgatherv): whatever the
            # variants below do on hardware they have never seen -- fail, or hang inside a collective -- there is a line to report.
            plain_ms = guarded("plain exchange, timed", lambda: measure(args.steps))
            log("plain exchange (all-gather after the product): %.4f ms per step over %d steps" % (plain_ms, args.steps))

            def vname(k):
                return "allreduce" if k[2] else "%d%s" % (k[0], "+p2p" if k[1] else "")

            def plain_line(note):
                return assemble(plain_ms, dict(dist_info, overlap_chunks=0, point_to_point=False, output_collective="exchange of the slices", exchange_variant="0",
                                               exchange_trials_ms={vname(k): v for k, v in trials.items()}, exchange_failures=failures, watchdog=note), False)

            def report_plain(stage):  # host-only: numbers measured earlier, no torch / HIP call (a collective may be hanging on the device)
                if rank == 0:
                    emit(plain_line("stage '%s' did not finish within %g s: the timing of the plain exchange, measured before it, is reported" % (stage, stage_limit)))
            trials, failures = {}, {}
            wd.set_fallback(report_plain)
            if rank == 0 and os.environ.get("HMX_BENCH_SIDE_FILE"):  # the parent relays this if the run dies before its own line (spawn_ranks_if_needed)
                try:
                    with open(os.environ["HMX_BENCH_SIDE_FILE"], "w") as fh:
                        fh.write(json.dumps(json_clean(plain_line("side file: written when the plain exchange had been timed; the run ended before its own line"))) + "\n")
                except OSError as e:
                    log("side file not written: %r" % (e,))
            pin_c, pin_p = os.environ.get("HMX_DIST_OVERLAP"), os.environ.get("HMX_DIST_P2P")
            chunk_choices = [int(pin_c)] if pin_c is not None else [0, 2, 4]
            p2p_choices = [bool(int(pin_p))] if pin_p is not None else [False, True]


            def agree(ok):  # a variant counts only when it worked on EVERY rank (over the torch process group: not the communicator under test)
                flag = torch.tensor([1 if ok else 0], device=flag_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return int(flag.item()) == 1

            def time_variant(key, name):
                def body():
                    err = None
                    try:
                        good = reproduces()
                    except Exception as e:  # noqa: BLE001 -- an error code out of RCCL: this variant is out, the run goes on
                        good, err = False, repr(e)
                    if not agree(err is None):
                        failures[name] = err or "failed on another rank"
                        log("exchange variant %s failed (%s): skipped" % (name, failures[name]))
                        return
                    if not good:
                        failures[name] = "result differs"
                        log("exchange variant %s does not reproduce the result: skipped" % name)
                        return
                    fence()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        step()
                    fence()
                    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=flag_dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    trials[key] = float(tt.item()) / 10 * 1e3
                guarded("exchange variant " + name, body)

            for p2p in p2p_choices:
                try:
                    native.set_point_to_point(p2p)
                    p2p_ok = 1
                except Exception as e:  # noqa: BLE001 -- no ncclSend / ncclRecv in this communicator
                    log("point-to-point exchange unavailable (%r)" % (e,))
                    p2p_ok = 0
                agreed = torch.tensor([p2p_ok], device=flag_dev)
                dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
                if int(agreed.item()) == 0:
                    native.set_point_to_point(False)
                    continue
                for chunks in chunk_choices:
                    def setup(chunks=chunks):
                        used = native.set_overlap(chunks, like=out)
                        if mu > 1 and chunks > 1:
                            # several right-hand sides: the row chunks of THEIR layout are agreed on inside the first product (the expanded
                            # view of a compact symmetric operator can be chunked where its fused single-vector product cannot)
                            step()
                            torch.cuda.synchronize()
                            used = native.overlap_chunks_multi()
                        return used
                    used = guarded("set-up of exchange variant %d%s" % (chunks, "+p2p" if p2p else ""), setup)
                    if chunks > 1 and used != chunks:
                        continue  # some rank's operator cannot be chunked
                    time_variant((chunks, p2p, False), "%d%s" % (chunks, "+p2p" if p2p else ""))
            # the north star's wording: ncclAllReduce of the zero-padded output vector (single vector; p times the bytes)
            if mu == 1 and os.environ.get("HMX_DIST_ALLREDUCE", "1") != "0":
                native.set_point_to_point(False)
                native.set_overlap(0, like=out)
                native.set_output_collective(True)
                time_variant((0, False, True), "allreduce")
                native.set_output_collective(False)
            if os.environ.get("HMX_DIST_ALLREDUCE") == "only" and (0, False, True) in trials:
                trials = {(0, False, True): trials[(0, False, True)]}
            best = min(trials, key=trials.get) if trials else (0, False, False)  # nothing could be timed (pinned to what no rank supports): the plain exchange

            def select():
                native.set_point_to_point(best[1])
                native.set_overlap(best[0], like=out)
                native.set_output_collective(best[2])
            guarded("selection of exchange variant " + vname(best), select)
            dist_info.update(overlap_chunks=best[0], point_to_point=bool(best[1]), output_collective="allreduce" if best[2] else "exchange of the slices",
                             exchange_variant=vname(best), exchange_trials_ms={vname(k): v for k, v in trials.items()}, exchange_failures=failures or None,
                             plain_exchange_ms=plain_ms)
            log("output exchange variants (ms per step; chunks of the expand stage, +p2p = pairwise send/recv, allreduce = zero-padded vector): %s -> %s" % (dist_info["exchange_trials_ms"], vname(best)))
        del ref
    if native is None and part and mu == 1 and args.trans == "N" and not os.environ.get("HMX_BENCH_NO_GRAPH"):
        eager_step = step
        eager_step()
        torch.cuda.synchronize()
        y_ref = y.clone()
        gp = D.GraphedGlobalToGlobalProduct(A, xin, y)
        y.zero_()
        gp()
        torch.cuda.synchronize()
        ok = torch.tensor([1 if (gp.graph is not None and torch.equal(y, y_ref)) else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # informational: ranks may differ, both paths issue the same collective
        graphed = bool(int(ok.item()))
        if torch.equal(y, y_ref):
            step = gp
        else:
            log("graphed product does not reproduce the eager result: timing eager launches")
    ms_per_step = guarded("timed region", lambda: measure(args.steps))
    if args.dump_product and mu == 1 and args.trans == "N":
        from oracle.oracle import hashed_vector  # (the checker's input generator only: a fixed, documented sequence)
        keep = xin.clone()
        xin.copy_(torch.from_numpy(hashed_vector(n, 1).astype(np_dt)).to(dev))

        def dumped():
            (y if use_dist else y_loc).zero_()
            step()
            torch.cuda.synchronize()
        guarded("product for --dump-product", dumped)
        if rank == 0:
            np.savez(args.dump_product, y=(y if use_dist else y_loc).cpu().numpy(), perm=np.asarray(T.get_permutation(), dtype=np.int32), n_gpus=world,
                     variant=str(dist_info.get("exchange_variant")))
        xin.copy_(keep)
    if not use_dist:
        roofline, kern_ms, nprof = local_roofline()

    # per rank: kernel time of the local product (HIP events), its algorithmic bytes; what the step adds on top is the exposed exchange
    if use_dist:
        mine = torch.tensor([sum(kern_ms.values()), esz * (st["cgen_dense"] + st["cgen_lowrank"] + n + H.nb_rows()) / 1e9], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dist_info.update(per_rank_local_ms=[float(t[0]) for t in allr], per_rank_GB=[float(t[1]) for t in allr])
        # what the exchange adds to a step, MEASURED: HIP events on the launch stream inside hmx_dist_matvec_global_to_global (start /
        # last local kernel done / whole result there), on the variant that was timed; maximum over the ranks of the per-rank means
        if native is not None and mu == 1:
            native.set_profiling(True)
            loc, exp = [], []
            for _ in range(nprof):
                step()
                a, b = native.last_exchange_ms()
                loc.append(a)
                exp.append(b)
            native.set_profiling(False)
            ev = torch.tensor([float(np.mean(loc)), float(np.mean(exp))], dtype=torch.float64, device=dev)
            dist.all_reduce(ev, op=dist.ReduceOp.MAX)
            dist_info.update(local_ms_events=float(ev[0]), exposed_exchange_ms=float(ev[1]), exposed_exchange_method="HIP events around the non-local part of the product (hmx_dist_last_exchange_ms), max over ranks")
        else:
            dist_info.update(exposed_exchange_ms=None, exposed_exchange_method="not measured on this path")
    # user numbering (permutations on the device) and host vectors (two PCIe copies per product): reported, never `value`
    extras = {}
    if not use_dist and mu == 1 and args.trans == "N" and not emu:
        try:
            xu, yu = xin.clone(), torch.zeros(n, dtype=t_dt, device=dev)
            for _ in range(3):
                hm.add_hmatrix_vector_product("N", 1.0, H, xu, 0.0, yu)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                hm.add_hmatrix_vector_product("N", 1.0, H, xu, 0.0, yu)
            torch.cuda.synchronize()
            extras["user_numbering_ms"] = (time.perf_counter() - t0) / 10 * 1e3
            xh, yh = xin.cpu().numpy(), np.zeros(n, dtype=np_dt)
            for _ in range(2):
                hm.internal_add_hmatrix_vector_product("N", 1.0, H, xh, 0.0, yh)
            t0 = time.perf_counter()
            for _ in range(5):
                hm.internal_add_hmatrix_vector_product("N", 1.0, H, xh, 0.0, yh)
            extras["host_vectors_pcie_inclusive_ms"] = (time.perf_counter() - t0) / 5 * 1e3
            if args.sym == "N" and world == 1:
                # the transposed product of the same operator: on the STORED data (mirrored column sums + owner-computes second sweep; no second
                # layout, bit-reproducible); the first call builds its index tables
                xt, yt = xin.clone(), torch.zeros(H.nb_cols(), dtype=t_dt, device=dev)
                t0 = time.perf_counter()
                hm.internal_add_hmatrix_vector_product("T", 1.0, H, xt, 0.0, yt)
                torch.cuda.synchronize()
                extras["transposed_first_call_ms"] = (time.perf_counter() - t0) * 1e3
                t0 = time.perf_counter()
                for _ in range(10):
                    hm.internal_add_hmatrix_vector_product("T", 1.0, H, xt, 0.0, yt)
                torch.cuda.synchronize()
                extras["transposed_ms"] = (time.perf_counter() - t0) / 10 * 1e3
                extras["transposed_tables_GB"] = H.stats()["transposed_bytes"] / 1e9
        except Exception as e:  # noqa: BLE001
            extras["error"] = repr(e)
    if not use_dist:
        compress = compress_info()

    compress["generator"] = args.generator
    if args.generator == "callback":
        compress["callback_build_s"] = t_build
        compress["callback_threads"] = args.callback_threads if args.callback_threads > 0 else min(64, hm.lib().hmx_host_cores())
    elif rank == 0 and world == 1 and not emu and not args.no_callback_build and os.path.exists(os.path.join(ROOT, "examples", "libhostgen.so")):
        # the literal drop-in route next to the device-kernel build: the SAME operator built again from a user's host generator (compiled code
        # libhmx knows nothing about, called on all cores; lock-step ACA and every product on the device), timed, compared, dropped
        try:
            t0 = time.time()
            Hc = tb.build(host_generator(hm, x, np_dt, cplx, args.sym == "H", args.callback_threads), T, T, brank, brank, device=local_rank, dtype=np_dt)
            torch.cuda.synchronize()
            compress["callback_build_s"] = time.time() - t0
            compress["callback_threads"] = args.callback_threads if args.callback_threads > 0 else min(64, hm.lib().hmx_host_cores())
            compress["callback_ranks_equal_device_build"] = bool(np.array_equal(Hc.leaf_table(), H.leaf_table()))
            stc = Hc.stats()
            compress["callback_entries_per_s"] = (stc["cgen_dense"] + stc["cgen_lowrank"]) / compress["callback_build_s"]
            del Hc
        except Exception as e:  # noqa: BLE001 -- a reported extra, never the product path
            compress["callback_build_s"] = None
            compress["callback_error"] = repr(e)

    out = assemble(ms_per_step, dist_info, graphed)
    if extras:
        out["other_entry_points"] = extras
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.dtype == "f64":
        try:
            cb, y_cpu, cut = cpu_baseline(H, T, args.cpu_sample_frac, log)
            # the same sample through the engine: parity on the slab while we are at it
            xh = torch.from_numpy(__import__("oracle.oracle", fromlist=["x"]).hashed_vector(n, 1)).to(dev)
            yh = torch.zeros(H.nb_rows(), dtype=torch.float64, device=dev)
            hm.internal_add_hmatrix_vector_product("N", 1.0, H, xh, 0.0, yh)
            err = float(np.linalg.norm(yh[:cut].cpu().numpy() - y_cpu) / np.linalg.norm(y_cpu))
            cb["rel_err_engine_vs_cpu_on_sample"] = err
            out["cpu_baseline"] = cb
            # htool itself on the same configuration: the `cpu_baseline` of kind "reference" when its binary is here; the port stays next to it
            ref = None if args.no_reference or args.sym != "N" else reference_baseline(log, n, args.geom, args.eps, args.eta, args.leaf, d, hm.lib().hmx_host_cores())
            if ref is not None:
                out["cpu_baseline_port"] = cb
                out["cpu_baseline"] = ref
        except Exception as e:  # the baseline is a reported number, never the product path
            out["cpu_baseline"] = dict(value=None, unit="GB/s", cores=0, kind="port", sample="failed: %r" % (e,))
    if use_dist:
        # the measured result must survive whatever the teardown does: bounded wait, then on regardless
        import threading
        th = threading.Thread(target=dist.destroy_process_group, daemon=True)
        th.start()
        th.join(60)
        if th.is_alive():
            log("destroy_process_group did not return within 60 s: continuing without it")
    if rank == 0 and world > 1 and not args.no_cpu_baseline and not args.no_reference and args.dtype == "f64" and args.sym == "N":
        # next to the multi-GPU numbers: htool's own MPI + OpenMP path on this box's host cores (the other ranks have finished)
        try:
            ref = reference_mpi_baseline(log, world, n, args.geom, args.eps, args.eta, args.leaf, d, hm.lib().hmx_host_cores())
        except Exception as e:  # noqa: BLE001 -- a reported extra: its failure must not cost the GPU result
            log("reference-mpi baseline failed: %r" % (e,))
            ref = None
        if ref is not None:
            out["cpu_baseline"] = ref
    if rank == 0:
        emit(out)


if __name__ == "__main__":
    main()
