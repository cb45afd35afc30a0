#!/usr/bin/env python3
"""Rewrites the rows of the round's bench files in profiles/README.md from the stored JSON files (after tools/store_profiles.py <round>)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def J(f):
    return json.load(open(os.path.join(P, f + ".json")))


def tot(p, key):
    for k, v in p.items():
        if key in k:
            return (v["fetch_bytes_x2"] + v["write_bytes"]) / 1e9


def main():
    p = os.path.join(P, "README.md")
    s = open(p).read()
    i = s.index("| `r5_bench_n1e6{.json,_kernel_stats.csv,_under_rocprof.json,_pmc_summary.json}`")
    j = s.index("| `traffic.json` | PMC bytes per launch of the kernels above with the sha256")
    b, pm = J("r5_bench_n1e6"), J("r5_bench_n1e6_pmc_summary")
    w = b["compress"]["written_array_placement"]
    t = J("traffic")
    rows = []
    rows.append("| `r5_bench_n1e6{.json,_kernel_stats.csv,_under_rocprof.json,_pmc_summary.json}` | default: N = 1e6 fp64, partialACA ε = 1e-4 | %.3f ms = %.2f TB/s = %.3f (2.82–2.85 over the boxes of the pool); `expand_kernel<4>` %.3f ms (%.3f), PMC %.2f GB = %.2f× algorithmic; `reduce_kernel` %.3f ms, PMC %.2f GB (placement probe of its written array: stream alone %.0f GB/s, first fit %.0f, chosen %.0f); compression kernels %.0f ms, pack kernels %.0f ms |" % (
        b["ms_per_step"], b["value"] / 1e3, b["value"] / 8000, b["roofline"]["kernels_ms"]["expand_kernel"], b["roofline"]["frac"], tot(pm, "expand_kernel"), tot(pm, "expand_kernel") / 11.46,
        b["roofline"]["kernels_ms"]["reduce_kernel"], tot(pm, "reduce_kernel"), w["stream_alone_GBps"], w["first_fit_GBps"], w["chosen_GBps"], 1e3 * b["compress"]["aca_kernels_s"], 1e3 * b["compress"]["pack_kernels_s"]))
    b = J("r5_bench_n1e6_sym")
    rows.append("| `r5_bench_n1e6_sym*` | `--sym S` | %.3f ms (%.2f by `B_alg`); %.2f GB moved = %.2f× the triangle |" % (b["ms_per_step"], b["value"] / 8000, t["sym_product_hbm_bytes_total"] / 1e9, t["sym_product_hbm_bytes_total"] / 1e9 / 9.53))
    b = J("r5_bench_n1e6_mu16")
    rows.append("| `r5_bench_n1e6_mu16*` | `--mu 16` | %.3f ms (%.3f); `expand_mfma16s` %.2f ms / %.2f GB moved, `reduce_mfma16s` %.2f ms / %.2f GB moved |" % (
        b["ms_per_step"], b["value"] / 8000, b["roofline"]["kernels_ms"]["expand_mfma16s_kernel"], t["mu16_expand_kernel_hbm_bytes_per_launch"] / 1e9, b["roofline"]["kernels_ms"]["reduce_mfma16s_kernel"],
        t["mu16_reduce_kernel_hbm_bytes_per_launch"] / 1e9))
    b = J("r5_bench_n1e6_transT")
    rows.append("| `r5_bench_n1e6_transT*` | `--trans T` (stored data) | %.3f ms (%.2f); %.1f GB moved = %.2f× |" % (b["ms_per_step"], b["value"] / 8000, t["transT_product_hbm_bytes_total"] / 1e9, t["transT_product_hbm_bytes_total"] / 1e9 / 18.56))
    b = J("r5_bench_n1e6_sym_mu16_stored_triangle")
    k = b["roofline"]["kernels_ms"]
    rows.append("| `r5_bench_n1e6_sym_mu16_stored_triangle*` | `--sym S --mu 16 --option sym_multi_rhs=1` | %.3f ms (round 4: 4.16): reduce %.3f, E pass %.3f, folds %.3f, second R pass %.3f |" % (
        b["ms_per_step"], k["reduce_mfma16s_kernel"], k["expand_sym_mfma16_kernel"], k["combine_sym_mu_kernel"], k["rowsym_mfma16_kernel"]))
    b = J("r5_bench_n1e6_sym_mu16_expanded_view")
    rows.append("| `r5_bench_n1e6_sym_mu16_expanded_view*` | `--sym S --mu 16 --option sym_multi_rhs=0` | %.3f ms on 9.4 + 18.6 GB |" % b["ms_per_step"])
    b, pm = J("r5_bench_c5_rank3of8"), J("r5_bench_c5_rank3of8_pmc_summary")
    mv = sum((v["fetch_bytes_x2"] + v["write_bytes"]) for kk, v in pm.items() if "read16" not in kk and "copy16" not in kk) / 1e9
    rows.append("| `r5_bench_c5_rank3of8*` | `--n 4000000 --sym S --dtype f32 --eps 1e-6 --mu 16 --emulate-world 8 --emulate-rank 3` (configs[4], one rank's operator) | %.3f ms for 7.84 GB algorithmic (%.2f): the product runs on the expanded view of the rank's operator (10.6 GB of streams), %.1f GB moved by the counters |" % (
        b["ms_per_step"], b["value"] / 8000, mv))
    s = s[:i] + "\n".join(rows) + "\n" + s[j:]
    open(p, "w").write(s)


if __name__ == "__main__":
    main()
