#!/usr/bin/env python3
"""Runs ON THE GPU BOX: builds ONE operator (bench.py's flags for geometry / size / symmetry / coefficient type / partition) and times
products of it under several settings of the per-operator options (hmx_hmatrix_set_option) -- one build, many A/B variants, all on the same
box and the same buffers.  Prints one line per variant: ms per product (HIP events around `steps` back-to-back products), the per-kernel
times of the profiled run, and the operator's stream statistics.

    python3 tools/probe.py --n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3 \
        --variant default --variant matrix_cores=0 --variant sym_multi_rhs=1
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--geom", default="ellipse")
    ap.add_argument("--eps", type=float, default=1e-4)
    ap.add_argument("--eta", type=float, default=10.0)
    ap.add_argument("--leaf", type=int, default=100)
    ap.add_argument("--sym", default="N")
    ap.add_argument("--trans", default="N")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32", "z64", "c32"])
    ap.add_argument("--mu", type=int, default=1)
    ap.add_argument("--emulate-world", type=int, default=0)
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--build-option", action="append", default=[], help="name=value, set before the build (layout / build options)")
    ap.add_argument("--variant", action="append", default=[], help="'default' or comma-separated name=value product options")
    ap.add_argument("--check", action="store_true", help="compare every variant's result with the first variant's")
    ap.add_argument("--spacer-gb", type=float, default=0.0, help="with --move-buffers: allocate (and keep) this much between two placements of X / Y")
    ap.add_argument("--reserve-gb", type=float, default=0.0, help="size of the slab reserved before the build (default: bench.py's rule, 64 KB per point, <= 60 %% of the free memory)")
    ap.add_argument("--move-buffers", action="store_true", help="before every variant after the first: new X / Y tensors and a product with more right-hand sides (the "
                    "operator's work area is then allocated again, elsewhere) -- how much of a difference is the placement of the buffers")
    args = ap.parse_args()

    import torch
    import bench
    import htool_amd as hm

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = args.n
    x = hm.create_geometry(args.geom, n)
    ctb = hm.ClusterTreeBuilder()
    ctb.set_maximal_leaf_size(args.leaf)
    emu = args.emulate_world
    T = ctb.create_cluster_tree(n, 3, x, 2, emu if emu else 2)
    tb = hm.HMatrixTreeBuilder(args.eps, args.eta, args.sym, "L" if args.sym != "N" else "N")
    tb.set_low_rank_generator("partialACA" if args.sym == "N" else "sympartialACA")
    d = bench.minimal_depth(n)
    tb.set_minimal_target_depth(d)
    tb.set_minimal_source_depth(d)
    for kv in args.build_option:
        k, v = kv.split("=")
        tb.set_option(k, float(v))
    cplx = args.dtype in ("z64", "c32")
    np_dt = {"f64": np.float64, "f32": np.float32, "z64": np.complex128, "c32": np.complex64}[args.dtype]
    t_dt = {"f64": torch.float64, "f32": torch.float32, "z64": torch.complex128, "c32": torch.complex64}[args.dtype]
    hm.lib().hmx_device_init(0)
    free_b, _ = torch.cuda.mem_get_info(0)
    hm.lib().hmx_device_reserve(0, int(args.reserve_gb * 1e9) if args.reserve_gb > 0 else int(min(0.6 * free_b, 65536.0 * n * np.dtype(np_dt).itemsize / 8 / max(1, emu))))
    gen = hm.InvDistGenerator(3, x, x, 1e-5, 1.0, 1.0, 1.0 if cplx else 0.0, args.sym == "H")
    brank = args.emulate_rank if emu else -1
    t0 = time.time()
    H = tb.build(gen, T, T, brank, brank, device=0, dtype=np_dt)
    torch.cuda.synchronize()
    st = H.stats()
    esz = np.dtype(np_dt).itemsize
    print("[probe] build %.2fs: %d dense + %d low-rank leaves, rank %d/%.2f/%d; rows %d; E %.3f GB, R %.3f GB, streams %.3f GB, a_total %d (x mu x esz = %.3f GB), C_gen %.3f GB" % (
        time.time() - t0, st["n_dense"], st["n_lowrank"], st["rank_min"], st["rank_mean"], st["rank_max"], H.nb_rows(), st["expand_coeffs"] * esz / 1e9,
        st["reduce_coeffs"] * esz / 1e9, st["stream_bytes"] / 1e9, st["a_total"], st["a_total"] * args.mu * esz / 1e9, (st["cgen_dense"] + st["cgen_lowrank"]) * esz / 1e9), flush=True)
    mu = args.mu
    rng = np.random.default_rng(1)
    nin, nout = (n, H.nb_rows()) if args.trans == "N" else (H.nb_rows(), n)
    if mu > 1:
        X = torch.from_numpy(rng.random((nin, mu)).astype(np_dt)).to(dev)
        Y = torch.zeros((nout, mu), dtype=t_dt, device=dev)
    else:
        X = torch.from_numpy(rng.random(nin).astype(np_dt)).to(dev)
        Y = torch.zeros(nout, dtype=t_dt, device=dev)

    def product():
        if mu > 1:
            hm.internal_add_hmatrix_matrix_product_row_major(args.trans, 1.0, H, X, 0.0, Y, mu)
        else:
            hm.internal_add_hmatrix_vector_product(args.trans, 1.0, H, X, 0.0, Y)

    ref = None

    def ref_done(m):
        return m > 0
    b_alg = esz * (st["cgen_dense"] + st["cgen_lowrank"] + mu * (n + H.nb_rows()))
    moves = 0
    spacers = []
    for var in args.variant or ["default"]:
        if args.move_buffers and ref_done(moves):
            if args.spacer_gb > 0:  # what torch hands out next lies `spacer_gb` further on (the spacers stay: physical memory is handed out in order)
                spacers.append(torch.empty(int(args.spacer_gb * (1 << 30)), dtype=torch.uint8, device=dev))
            keep = [torch.empty((3 + 5 * moves) << 20, dtype=torch.uint8, device=dev)]  # shifts what torch hands out next
            X, Y = X.clone(), torch.zeros_like(Y)
            if mu > 1:
                mu2 = mu + 16 * (moves + 1)
                X2 = torch.zeros((nin, mu2), dtype=t_dt, device=dev)
                Y2 = torch.zeros((nout, mu2), dtype=t_dt, device=dev)
                hm.internal_add_hmatrix_matrix_product_row_major(args.trans, 1.0, H, X2, 0.0, Y2, mu2)
                del X2, Y2
            del keep
        moves += 1
        opts = {} if var == "default" else dict(kv.split("=") for kv in var.split(","))
        saved = {k: H.get_option(k) for k in opts}
        for k, v in opts.items():
            H.set_option(k, float(v))
        for _ in range(3):
            product()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            product()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        H.set_profiling(True)
        acc = {}
        for _ in range(5):
            product()
            for name, t in H.last_kernel_times():
                acc.setdefault(name, []).append(t)
        H.set_profiling(False)
        kern = {k: round(float(np.mean(v)), 3) for k, v in acc.items()}
        st2 = H.stats()
        line = dict(variant=var, ms=round(ms, 3), GBps_alg=round(b_alg / ms / 1e6, 1), frac=round(b_alg / ms / 1e6 / 8000, 3), kernels_ms=kern,
                    expanded_GB=round(st2["expanded_bytes"] / 1e9, 2), transposed_GB=round(st2["transposed_bytes"] / 1e9, 2))
        if args.check:
            out = Y.detach().cpu().numpy().copy()
            if ref is None:
                ref = out
            line["rel_diff_vs_first"] = float(np.linalg.norm(out - ref) / max(np.linalg.norm(ref), 1e-300))
        print("[probe] " + json.dumps(line), flush=True)
        for k, v in saved.items():
            H.set_option(k, v)


if __name__ == "__main__":
    main()
