#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the REAL reference (dev container only).

Runs oracle/_ref/ref_driver (oracle/ref/ref_driver.cpp compiled against the htool headers where they
lie under /root/reference/include + the image's MKL; `make -C oracle ref`) on a list of small cases and
stores inputs + expected outputs as compressed .npz fixtures.  The fixtures are DATA (seeded inputs,
permutations, cluster tables, leaf lists, ranks, a few U/V/dense payloads, matvec results); no reference
source text is stored.  The reference's own tests hold no golden vectors for this path (SURVEY.md section 4:
random inputs, tolerance checks against a dense product), so outputs of the reference itself are the pin.

Usage: python tests/golden/make_golden.py [case ...]       (needs /root/reference; not run on the GPU box)
       python tests/golden/make_golden.py full [case ...]  BASELINE's own sizes (N=1e5, 1e6): hashes + sampled products
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.oracle import read_dump  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")

# name -> (mode, params).  Parameters mirror ref_driver's key=value options.
CASES = {
    # plumbing config C1 analogue (BASELINE.json configs[0]) at fixture size; keeps coords to pin geometry
    "ball_n2000_partial": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-3, eta=10, compressor="partialACA", keep_coords=1)),
    "ellipse_n3000_partial": ("hmat", dict(n=3000, geom="ellipse", leaf=100, eps=1e-4, eta=10, compressor="partialACA", keep_coords=1)),
    # shipped-example style: symmetric storage, default compressor (sympartialACA), Partitioning_N
    "ellipse_n3000_symL_default": ("hmat", dict(n=3000, geom="ellipse", leaf=100, eps=1e-4, eta=10, sym="S", uplo="L", compressor="default", partitioning="n_pca_regular")),
    "ellipse_n3000_symU_sympartial": ("hmat", dict(n=3000, geom="ellipse", leaf=100, eps=1e-4, eta=10, sym="S", uplo="U", compressor="sympartialACA")),
    "ball_n2000_symL_eta3": ("hmat", dict(n=2000, geom="ball", leaf=40, eps=1e-3, eta=3, sym="S", uplo="L", compressor="sympartialACA")),
    # single-partition quirk (root with exactly one child)
    "disk_n1500_p1": ("hmat", dict(n=1500, geom="disk", leaf=60, partitions=1, eps=1e-4, compressor="partialACA")),
    # minimal block depth (SURVEY.md B-1 workaround used at N=1e6)
    "ellipse_n4000_mindepth3": ("hmat", dict(n=4000, geom="ellipse", leaf=100, eps=1e-4, mindepth=3, compressor="partialACA")),
    # rectangular, different target/source geometry
    "rect_ball1500_disk1000": ("hmat", dict(n=1500, nsrc=1000, geom="ball", sgeom="disk", sz=2.5, leaf=60, eps=1e-4, compressor="partialACA")),
    # clustering strategies
    "ball_n2000_bbox_regular": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-3, partitioning="bbox_regular", dump_blocks=0)),
    "ball_n2000_pca_geometric": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-3, partitioning="pca_geometric", dump_blocks=0)),
    "ball_n2000_bbox_geometric": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-3, partitioning="bbox_geometric", dump_blocks=0)),
    "ball_n2000_n_pca_c4": ("hmat", dict(n=2000, geom="ball", leaf=50, children=4, partitions=4, eps=1e-3, partitioning="n_pca_regular", dump_blocks=0)),
    "ball_n2000_n_bbox_c8": ("hmat", dict(n=2000, geom="ball", leaf=30, children=8, partitions=8, eps=1e-3, partitioning="n_bbox_regular", dump_blocks=0)),
    "ellipse_n2000_c3_p3": ("hmat", dict(n=2000, geom="ellipse", leaf=50, children=3, partitions=3, eps=1e-3, dump_blocks=0)),
    # user-given partitions (create_cluster_tree_from_{global,local}_partition) and complete trees (set_is_complete)
    "ball_n2000_given_global_p3": ("hmat", dict(n=2000, geom="ball", leaf=50, partitions=3, eps=1e-3, given="global", dump_blocks=0)),
    "ellipse_n2000_given_local_p4": ("hmat", dict(n=2000, geom="ellipse", leaf=50, partitions=4, eps=1e-3, given="local", dump_blocks=0)),
    "ball_n2000_given_global_p4_rank2": ("hmat", dict(n=2000, geom="ball", leaf=50, partitions=4, rank=2, eps=1e-3, given="global", dump_blocks=0)),
    "ball_n2000_complete": ("hmat", dict(n=2000, geom="ball", leaf=60, eps=1e-3, complete=1, dump_blocks=0)),
    "ellipse_n1500_complete_c3": ("hmat", dict(n=1500, geom="ellipse", leaf=40, children=3, partitions=3, eps=1e-3, complete=1, dump_blocks=0)),
    # 2-D point cloud (solve_EVP_2 branch), non-consistent block tree, near-full-rank accuracy
    "disk2d_n2000": ("hmat", dict(n=2000, geom="disk2d", leaf=50, eps=1e-4, eta=10, compressor="partialACA")),
    "disk2d_n2000_bbox_symL": ("hmat", dict(n=2000, geom="disk2d", leaf=50, eps=1e-4, eta=5, sym="S", uplo="L", compressor="sympartialACA", partitioning="bbox_regular")),
    "rect_ball1500_disk1000_nonconsistent": ("hmat", dict(n=1500, nsrc=1000, geom="ball", sgeom="disk", sz=2.5, leaf=60, eps=1e-4, compressor="partialACA", consistent=0)),
    "ball_n1500_eps1e-12": ("hmat", dict(n=1500, geom="ball", leaf=50, eps=1e-12, eta=10, compressor="partialACA", dump_blocks=1)),
    "ball_n1500_eps1e-8": ("hmat", dict(n=1500, geom="ball", leaf=100, eps=1e-8, eta=10, compressor="partialACA", dump_blocks=1)),
    "ball_n300_small": ("hmat", dict(n=300, geom="ball", leaf=100, eps=1e-4, compressor="partialACA", partitions=4, dump_blocks=1)),
    # fp32 coefficients, fp64 geometry: htool's HMatrix<float,double> (BASELINE config 5 precision)
    "ellipse_n3000_f32_partial": ("hmat", dict(n=3000, geom="ellipse", leaf=100, eps=1e-4, eta=10, compressor="partialACA", prec="f32")),
    "ellipse_n3000_f32_symL_eps1e-6": ("hmat", dict(n=3000, geom="ellipse", leaf=100, eps=1e-6, eta=10, sym="S", uplo="L", compressor="sympartialACA", prec="f32")),
    "ball_n2000_f32_p2_rank1": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-3, partitions=2, rank=1, compressor="partialACA", prec="f32", dump_blocks=1)),
    # complex coefficients: HMatrix<std::complex<double>> / <std::complex<float>>, generator
    # (cre + i cim sgn)/(delta + scale |x-y|) -- complex symmetric form of testing/generator_test.hpp:163-196, Hermitian form :198-205
    "ball_n2000_z64_partial": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-4, compressor="partialACA", prec="z64", dump_blocks=2)),
    "ellipse_n3000_z64_symL": ("hmat", dict(n=3000, geom="ellipse", leaf=100, eps=1e-4, sym="S", uplo="L", compressor="sympartialACA", prec="z64", cre=0.3, cim=-1.2, dump_blocks=2)),
    "ball_n2000_z64_hermU": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-4, sym="H", uplo="U", compressor="sympartialACA", prec="z64", dump_blocks=2)),
    "ball_n2000_c32_hermL": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-3, sym="H", uplo="L", compressor="default", prec="c32", dump_blocks=1)),
    "ellipse_n3000_c32_partial": ("hmat", dict(n=3000, geom="ellipse", leaf=100, eps=1e-4, compressor="partialACA", prec="c32", dump_blocks=1)),
    "ball_n2000_z64_p2_rank1": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-3, partitions=2, rank=1, compressor="partialACA", prec="z64", dump_blocks=1)),
    "ball_n2000_z64_p2_hermL_rank0": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-3, partitions=2, rank=0, sym="H", uplo="L", compressor="sympartialACA", prec="z64", dump_blocks=1)),
    "rect_ball1500_disk1000_z64": ("hmat", dict(n=1500, nsrc=1000, geom="ball", sgeom="disk", sz=2.5, leaf=60, eps=1e-4, compressor="partialACA", prec="z64", dump_blocks=1)),
    "ball_n1200_z64_fullACA": ("hmat", dict(n=1200, geom="ball", leaf=50, eps=1e-4, compressor="fullACA", prec="z64", dump_blocks=1)),
    "ball_n1200_z64_SVD": ("hmat", dict(n=1200, geom="ball", leaf=50, eps=1e-4, compressor="SVD", prec="z64", dump_blocks=1)),
    "ball_n2000_z64_hermU_recompressed": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-4, sym="H", uplo="U", compressor="sympartialACA", prec="z64", recompress=1, dump_blocks=1)),
    "ellipse_n3000_c32_recompressed": ("hmat", dict(n=3000, geom="ellipse", leaf=100, eps=1e-4, compressor="partialACA", prec="c32", recompress=1, dump_blocks=1)),
    # the other device kernel families (include/hmx.h: HMX_KERNEL_HELMHOLTZ exp(i k r) / (delta + scale r), HMX_KERNEL_LAPLACE_SL
    # 1 / (4 pi (delta + r))), written as user generators in oracle/ref/ref_driver.cpp (kernel=..., wavenumber=...)
    "ball_n2000_z64_helmholtz": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-4, compressor="partialACA", prec="z64", kernel="helmholtz", wavenumber=5.0, delta=1e-5, scale=12.566370614359172, dump_blocks=2)),
    "ellipse_n3000_z64_helmholtz_symL": ("hmat", dict(n=3000, geom="ellipse", leaf=100, eps=1e-4, sym="S", uplo="L", compressor="sympartialACA", prec="z64", kernel="helmholtz", wavenumber=3.0, delta=1e-5, scale=12.566370614359172, dump_blocks=2)),
    "ellipse_n3000_helmholtz_real_symL": ("hmat", dict(n=3000, geom="ellipse", leaf=100, eps=1e-4, sym="S", uplo="L", compressor="sympartialACA", kernel="helmholtz", wavenumber=2.0, delta=1e-5, scale=12.566370614359172, dump_blocks=2)),
    "ball_n2000_laplace": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-4, compressor="partialACA", kernel="laplace", delta=1e-5, dump_blocks=2)),
    "ball_n1200_z64_reqrank5": ("hmat", dict(n=1200, geom="ball", leaf=50, eps=1e-4, reqrank=5, compressor="partialACA", prec="z64", dump_blocks=1)),
    # block-diagonal (local-to-local) operator rooted at the partition clusters: DefaultLocalApproximationBuilder
    "ellipse_n4000_p4_local2": ("hmat", dict(n=4000, geom="ellipse", leaf=100, partitions=4, local=2, eps=1e-4, compressor="partialACA", dump_blocks=1)),
    "ball_n2000_p2_local1_symL": ("hmat", dict(n=2000, geom="ball", leaf=50, partitions=2, local=1, eps=1e-3, sym="S", uplo="L", compressor="sympartialACA", dump_blocks=1)),
    # SVD recompression of the ACA output (recompression(hmatrix))
    "ellipse_n3000_recompressed": ("hmat", dict(n=3000, geom="ellipse", leaf=100, eps=1e-4, eta=10, compressor="partialACA", recompress=1)),
    "ball_n2000_symL_recompressed": ("hmat", dict(n=2000, geom="ball", leaf=50, eps=1e-3, eta=10, sym="S", uplo="L", compressor="sympartialACA", recompress=1)),
    # other compressors
    "ball_n1200_fullACA": ("hmat", dict(n=1200, geom="ball", leaf=50, eps=1e-4, compressor="fullACA", dump_blocks=2)),
    "ball_n1200_SVD": ("hmat", dict(n=1200, geom="ball", leaf=50, eps=1e-4, compressor="SVD", dump_blocks=2)),
    "ball_n1200_reqrank5": ("hmat", dict(n=1200, geom="ball", leaf=50, eps=1e-4, reqrank=5, compressor="partialACA", dump_blocks=2)),
    # row partition (DistributedOperator local blocks): p=4, every rank; symmetric with p=2
    **{"ellipse_n4000_p4_rank%d" % r: ("hmat", dict(n=4000, geom="ellipse", leaf=100, partitions=4, rank=r, eps=1e-4, compressor="partialACA", dump_blocks=1)) for r in range(4)},
    **{"ball_n2000_p2_symL_rank%d" % r: ("hmat", dict(n=2000, geom="ball", leaf=50, partitions=2, rank=r, eps=1e-3, sym="S", uplo="L", compressor="sympartialACA", dump_blocks=1)) for r in range(2)},
    **{"ball_n2000_p2_symU_rank%d" % r: ("hmat", dict(n=2000, geom="ball", leaf=50, partitions=2, rank=r, eps=1e-3, sym="S", uplo="U", compressor="sympartialACA", dump_blocks=1)) for r in range(2)},
    # the reference's on-disk formats (save_cluster_tree, read_cluster_tree -> save, save_leaves_with_rank, matrix_to_bytes):
    # the files the reference wrote, stored as bytes
    "io_ellipse_n1000_p2": ("io", dict(n=1000, geom="ellipse", leaf=50, eps=1e-3, partitions=2, compressor="partialACA")),
    "io_ball_n1200_c4_p4": ("io", dict(n=1200, geom="ball", leaf=30, children=4, partitions=4, eps=1e-3, partitioning="n_pca_regular", compressor="partialACA")),
    "io_disk2d_n800_symL_p1": ("io", dict(n=800, geom="disk2d", leaf=40, eps=1e-3, partitions=1, sym="S", uplo="L", compressor="sympartialACA")),
    "io_ball_n1500_p4_rank2": ("io", dict(n=1500, geom="ball", leaf=50, eps=1e-3, partitions=4, rank=2, compressor="partialACA")),
    # print_distributed_hmatrix_information of the reference run under MPI (oracle/_ref/dist_info, world = partitions): the
    # text rank 0 prints for the operators whose per-rank leaf tables are the *_p4_rank* / *_p2_symL_rank* fixtures above
    "distinfo_ellipse_n4000_p4": ("distinfo", dict(n=4000, geom="ellipse", leaf=100, partitions=4, eps=1e-4, compressor="partialACA")),
    "distinfo_ball_n2000_p2_symL": ("distinfo", dict(n=2000, geom="ball", leaf=50, partitions=2, eps=1e-3, sym="S", uplo="L", compressor="sympartialACA")),
    # EVERY product family of htool's DistributedOperator run under MPI (oracle/_ref/dist_products, world = partitions): vector /
    # row-major / column-major, global-to-global / local-to-local, user / partition numbering, sub product; one fixture holds the
    # outputs of all ranks (keys r<k>_<name>).  given=local: a cluster tree from a local partition, so that the local user numbering
    # of the add_*_local_to_local families exists (test_distributed_operator.hpp:520-537 builds its trees the same way)
    "distprod_ellipse_n2000_p2": ("distprod", dict(n=2000, geom="ellipse", leaf=50, partitions=2, eps=1e-6, compressor="partialACA", mu=5, given="local")),
    "distprod_ellipse_n2400_p4": ("distprod", dict(n=2400, geom="ellipse", leaf=50, partitions=4, eps=1e-6, compressor="partialACA", mu=3, given="local")),
    "distprod_ball_n1500_p2_symL": ("distprod", dict(n=1500, geom="ball", leaf=40, partitions=2, eps=1e-5, sym="S", uplo="L", compressor="sympartialACA", mu=5, given="local")),
    "distprod_ball_n1500_c3_p3": ("distprod", dict(n=1500, geom="ball", leaf=40, children=3, partitions=3, eps=1e-5, compressor="partialACA", mu=4)),
    "distprod_ellipse_n2000_p2_blockdiag": ("distprod", dict(n=2000, geom="ellipse", leaf=50, partitions=2, eps=1e-6, compressor="partialACA", mu=5, given="local", local=1)),
    "distprod_ball_n1200_p2_z64_hermL": ("distprod", dict(n=1200, geom="ball", leaf=40, partitions=2, eps=1e-5, sym="H", uplo="L", compressor="sympartialACA", mu=3, given="local", prec="z64")),
    # the reference's own compressor test block (500 x 100, two disks at distance d)
    **{"lrmat_d%d" % d: ("lrmat", dict(distance=d, eps=1e-4)) for d in (15, 20, 30, 40)},
}


# ---- BASELINE.json's own sizes, pinned to the reference ("full" mode) ----------------------------------------------------
# The reference builds these in this container (N=1e6 ellipse: minutes, 18.5 GB).  Whole outputs would be hundreds of MB,
# so a fixture keeps: sha256 of the permutation and of the leaf table's structure columns, the rank column, and the
# products at SAMPLE fixed rows (alpha=3, beta=2 and alpha=1, beta=0; row-major multi-RHS).  min depth at N=1e6 is the
# bench's (bench.minimal_depth: smallest d with N / 2^d <= 46340).
SAMPLE = 4096
FULL_CASES = {
    "full_ball_n100000": dict(n=100000, geom="ball", leaf=100, eps=1e-4, eta=10, compressor="partialACA"),
    "full_ellipse_n100000": dict(n=100000, geom="ellipse", leaf=100, eps=1e-4, eta=10, compressor="partialACA"),
    "full_ellipse_n1000000": dict(n=1000000, geom="ellipse", leaf=100, eps=1e-4, eta=10, compressor="partialACA", mindepth=5),
    "full_ellipse_n100000_symL": dict(n=100000, geom="ellipse", leaf=100, eps=1e-4, eta=10, sym="S", uplo="L", compressor="sympartialACA"),
    "full_ball_n100000_symL": dict(n=100000, geom="ball", leaf=100, eps=1e-4, eta=10, sym="S", uplo="L", compressor="sympartialACA"),
    # BASELINE config 5's shape at a size the reference builds here: fp32 coefficients, 'S','L', sympartialACA, eps=1e-6, 16 RHS
    "full_ellipse_n100000_f32_symL_mu16": dict(n=100000, geom="ellipse", leaf=100, eps=1e-6, eta=10, sym="S", uplo="L", compressor="sympartialACA", prec="f32", mu=16),
    # the reference's Hermitian test generator (sign-discontinuous imaginary part, testing/generator_test.hpp:186-205): blocks that cross the
    # discontinuity have ranks in the hundreds -- the case the ACA's pool growth and workgroup teams exist for
    "full_ellipse_n100000_z64_hermL": dict(n=100000, geom="ellipse", leaf=100, eps=1e-4, eta=10, sym="H", uplo="L", compressor="sympartialACA", prec="z64", mindepth=3),
}


# BASELINE configs[3]: the N=1e6 operator row-partitioned over 8 ranks (DistributedOperator: rank k owns the block rows of partition
# cluster k of the tree built with size_of_partition = 8).  One reference run per rank (target_partition_number = k): structure hash,
# ranks and the rank-local product at sampled local rows -- what rank k of htool's MPI run computes before the Allgatherv.
FULL_CASES.update({"full_ellipse_n1000000_p8_rank%d" % r: dict(n=1000000, geom="ellipse", leaf=100, eps=1e-4, eta=10, compressor="partialACA", mindepth=5, partitions=8, rank=r)
                   for r in range(8)})


# BASELINE configs[4] at its own size: N=4e6 fp32, 'S','L', sympartialACA, eps=1e-6, 16 right-hand sides, row-partitioned over 8 ranks
# (minimal depth 7: N / 2^7 <= 46340).  All eight per-rank operators as htool builds
# and multiplies them (5 GB each; ranks 0, 3, 7 since round 3, the other five since round 4).  And configs[0]'s literal workload: N=5000 points in the unit ball, eps=1e-3, partialACA.
FULL_CASES.update({"full_ellipse_n4000000_f32_symL_p8_rank%d" % r: dict(n=4000000, geom="ellipse", leaf=100, eps=1e-6, eta=10, sym="S", uplo="L", compressor="sympartialACA",
                                                                        prec="f32", mu=16, mindepth=7, partitions=8, rank=r) for r in range(8)})
FULL_CASES["full_ball_n5000_eps1e-3"] = dict(n=5000, geom="ball", leaf=100, eps=1e-3, eta=10, compressor="partialACA")


def sample_rows(n):
    """SAMPLE distinct fixed rows, spread over the whole range (closed form: reproducible in the test)."""
    return np.unique((np.arange(SAMPLE, dtype=np.int64) * 2654435761 + 12345) % n)


def make_full(only):
    import hashlib
    manifest = {}
    for name, params in FULL_CASES.items():
        if only and name not in only:
            continue
        with tempfile.NamedTemporaryFile(suffix=".bin", dir="/tmp") as tmp:
            cmd = [DRIVER, "hmat"] + ["%s=%s" % (k, v) for k, v in params.items()] + ["out=" + tmp.name, "par=1", "extra_ab=1", "dump_blocks=0"]
            print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd, env=dict(os.environ, OMP_NUM_THREADS=str(os.cpu_count())))
            d = read_dump(tmp.name)
        leaves = np.ascontiguousarray(d["leaves"], dtype=np.int32)
        structure = np.ascontiguousarray(leaves[:, [0, 1, 2, 3, 5]])
        rows = sample_rows(len(np.asarray(d["yN"])))  # local rows of a row-restricted operator
        out = dict(
            perm_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(d["t_perm"], dtype=np.int32).tobytes()).digest(), dtype=np.uint8),
            structure_sha256=np.frombuffer(hashlib.sha256(structure.tobytes()).digest(), dtype=np.uint8),
            nleaves=np.int64(len(leaves)),
            ranks=leaves[:, 4].astype(np.int16),
            rows=rows,
            yN=np.asarray(d["yN"])[rows],
            yN_a1b0=np.asarray(d["yN_a1b0"])[rows],
            YNrm=np.asarray(d["YNrm"])[rows],
            alphabeta=d["alphabeta"],
            stats=d["stats"],
            rootinfo=d["rootinfo"],
        )
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        manifest[name] = dict(mode="full", **params)
        print(name, "leaves", len(leaves), "stats", d["stats"][:7], flush=True)
    return manifest


def main():
    if sys.argv[1:2] == ["full"]:
        if not os.path.exists(DRIVER):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        manifest = make_full(set(sys.argv[2:]))
        mpath = os.path.join(HERE, "manifest.json")
        with open(mpath) as f:
            old = json.load(f)
        old.update(manifest)
        with open(mpath, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
        return
    if not os.path.exists(DRIVER):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    manifest = {}
    only = set(sys.argv[1:])
    for name, (mode, params) in CASES.items():
        if only and name not in only:
            continue
        params = dict(params)
        keep_coords = params.pop("keep_coords", 0)
        if mode == "distinfo":
            with tempfile.TemporaryDirectory() as tmp:
                exe = os.path.join(ROOT, "oracle", "_ref", "dist_info")
                cmd = ["/opt/conda/bin/mpiexec", "-n", str(params["partitions"]), exe] + ["%s=%s" % (k, v) for k, v in params.items()] + ["out=" + tmp + "/info.txt"]
                print(" ".join(cmd))
                subprocess.check_call(cmd, env=dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "libs")))
                np.savez_compressed(os.path.join(HERE, name + ".npz"), information=np.fromfile(os.path.join(tmp, "info.txt"), dtype=np.uint8))
            manifest[name] = dict(mode=mode, **params)
            continue
        if mode == "distprod":
            with tempfile.TemporaryDirectory() as tmp:
                exe = os.path.join(ROOT, "oracle", "_ref", "dist_products")
                world = params["partitions"]
                cmd = ["/opt/conda/bin/mpiexec", "-n", str(world), exe] + ["%s=%s" % (k, v) for k, v in params.items()] + ["out=" + tmp + "/dp"]
                print(" ".join(cmd))
                subprocess.check_call(cmd, env=dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "libs"), OMP_NUM_THREADS="1"))
                d = {}
                for r in range(world):
                    dr = read_dump(tmp + "/dp.rank%d" % r)
                    for k, v in dr.items():
                        if params.get("prec") == "z64" and v.dtype == np.float64 and v.ndim >= 2 and v.shape[-1] == 2 and k not in ("alpha_beta",):
                            v = np.ascontiguousarray(v).view(np.complex128).reshape(v.shape[:-1])
                        if k in ("perm", "partition", "alpha_beta", "sub_offset_size"):
                            if r == 0:
                                d[k] = v
                        else:
                            d["r%d_%s" % (r, k)] = v
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
            manifest[name] = dict(mode=mode, **params)
            continue
        if mode == "io":
            with tempfile.TemporaryDirectory() as tmp:
                cmd = [DRIVER, "hmat"] + ["%s=%s" % (k, v) for k, v in params.items()] + ["out=" + tmp + "/d.bin", "dump_blocks=0", "save_prefix=" + tmp + "/f"]
                print(" ".join(cmd))
                subprocess.check_call(cmd)
                d = {}
                for key, fn in (("tree", "f_cluster_tree.csv"), ("properties", "f_cluster_tree_properties.csv"), ("reread_tree", "f_reread_cluster_tree.csv"),
                                ("reread_properties", "f_reread_cluster_tree_properties.csv"), ("leaves", "f_leaves.csv"), ("dense0", "f_dense0.bin"), ("information", "f_information.txt")):
                    d[key] = np.fromfile(os.path.join(tmp, fn), dtype=np.uint8)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
            manifest[name] = dict(mode=mode, **params)
            continue
        with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
            cmd = [DRIVER, mode] + ["%s=%s" % (k, v) for k, v in params.items()] + ["out=" + tmp.name]
            print(" ".join(cmd))
            subprocess.check_call(cmd)
            d = read_dump(tmp.name)
        if not keep_coords and mode == "hmat":
            d.pop("xt", None)
            d.pop("xs", None)
        if mode == "hmat":
            for k in ("x", "xT", "y0", "y0T"):  # closed-form inputs (oracle.hashed_vector), not stored
                d.pop(k, None)
        if params.get("prec") in ("z64", "c32"):  # float64 pairs -> complex128
            for k in list(d):
                if k[:2] in ("U_", "V_", "D_") or k[0] in "yY" or k.startswith("all"):
                    a = np.ascontiguousarray(d[k], dtype=np.float64)
                    d[k] = a.view(np.complex128).reshape(a.shape[:-1])
        if mode == "lrmat":
            d.pop("xt", None)
            d.pop("xs", None)
            if params["distance"] in (20, 30):  # keep ranks only for the middle distances
                d = {k: v for k, v in d.items() if k.endswith("_info") or k.endswith("perm")}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        manifest[name] = dict(mode=mode, **params)
    mpath = os.path.join(HERE, "manifest.json")
    if only and os.path.exists(mpath):  # partial regeneration: merge into the existing manifest
        with open(mpath) as f:
            old = json.load(f)
        old.update(manifest)
        manifest = old
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", len(manifest), "fixtures")


if __name__ == "__main__":
    main()
