"""Copies what tools/collect_profiles.sh left under gpurun_out/{rN_n1e6,rN_sym,rN_mu16,rN_transT,...} into profiles/ (tracked) and rewrites
profiles/traffic.json with the sha256 of the kernel sources the counters were measured on.  Run in the dev container right after the gpurun call:
    python3 tools/store_profiles.py r5"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

os.chdir(ROOT)
RN = sys.argv[1] if len(sys.argv) > 1 else "r5"
EXTRA = [(RN + "_sym_mu16_stored", RN + "_bench_n1e6_sym_mu16_stored_triangle"), (RN + "_sym_mu16_view", RN + "_bench_n1e6_sym_mu16_expanded_view"), (RN + "_c5_rank3", RN + "_bench_c5_rank3of8")]
for tag, name in [(RN + "_n1e6", RN + "_bench_n1e6"), (RN + "_sym", RN + "_bench_n1e6_sym"), (RN + "_mu16", RN + "_bench_n1e6_mu16"), (RN + "_transT", RN + "_bench_n1e6_transT")] + [e for e in EXTRA if os.path.isdir("gpurun_out/" + e[0])]:
    d = "gpurun_out/" + tag
    shutil.copy(d + "/kernel_stats.csv", "profiles/%s_kernel_stats.csv" % name)
    shutil.copy(d + "/under_rocprof.json", "profiles/%s_under_rocprof.json" % name)
    shutil.copy(d + "/bench.json", "profiles/%s.json" % name)
    f, w = json.load(open(d + "/pmc_FETCH_SIZE.json")), json.load(open(d + "/pmc_WRITE_SIZE.json"))
    summ = {}
    for k in sorted(set(f) | set(w)):
        if "pack" in k or "copy16" in k:
            continue
        fs, ws = f.get(k, {}).get("FETCH_SIZE", {}), w.get(k, {}).get("WRITE_SIZE", {})
        summ[k] = dict(FETCH_SIZE_KB_mean=fs.get("mean"), fetch_bytes_x2=2 * 1024 * fs["mean"] if fs else None, WRITE_SIZE_KB_mean=ws.get("mean"),
                       write_bytes=1024 * ws["mean"] if ws else None, launches=fs.get("n"))
    json.dump(summ, open("profiles/%s_pmc_summary.json" % name, "w"), indent=1, sort_keys=True)
n, s = json.load(open("profiles/%s_bench_n1e6_pmc_summary.json" % RN)), json.load(open("profiles/%s_bench_n1e6_sym_pmc_summary.json" % RN))
m16 = json.load(open("profiles/%s_bench_n1e6_mu16_pmc_summary.json" % RN))
tT = json.load(open("profiles/%s_bench_n1e6_transT_pmc_summary.json" % RN))


def opt(name):
    f = "profiles/%s_%s_pmc_summary.json" % (RN, name)
    return json.load(open(f)) if os.path.exists(f) else None


s16, c5 = opt("bench_n1e6_sym_mu16_stored_triangle"), opt("bench_c5_rank3of8")
# kernels of each workload's OWN product (the bench also multiplies in other shapes: the transposed product of `other_entry_points`, the
# single-vector product hmx_hmatrix_alloc_vector times for y -- their kernels have other names)
MU_N = ("reduce_mfma16s", "expand_mfma16s", "::combine_mu_kernel")
MU_SYM = ("reduce_mfma16s", "::combine_mu_kernel", "expand_sym_mfma16", "combine_list_mu", "rowsym_mfma16")
ONE_SYM = ("::reduce_kernel", "::combine_kernel", "::expand_sym_kernel", "::combine_list", "::rowsym_kernel")


def own(d, names):
    return sum(tot(v) for k, v in d.items() if any(x in k for x in names) and "read16" not in k)



def tot(x):
    return x["fetch_bytes_x2"] + x["write_bytes"]


def pick(d, sub):
    return next(v for k, v in d.items() if sub in k)


rec = dict(round=int(RN[1:]), kernel_sources_sha256=bench.kernel_sources_hash(),
           workload="bench.py N=1e6 ellipse eps=1e-4 (1 GPU): default (partialACA, 'N') and --sym S (sympartialACA, 'S','L', compact storage, fused product)",
           method="rocprofv3 --kernel-trace --pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE (tools/collect_profiles.sh, tools/pmc_summary.py); values in KB; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM): in the same runs the 8 GiB read16_kernel reports 4.194e6 KB = 1/2 of 8 GiB; WRITE_SIZE exact",
           product_hbm_bytes_total=sum(tot(v) for k, v in n.items() if any(k.endswith(x) or (x + "<") in k for x in ("::expand_kernel", "::reduce_kernel", "::combine_kernel"))),  # (the bench also times a transposed product: other kernels)
           expand_kernel_hbm_bytes_per_launch=tot(pick(n, "expand_kernel")), expand_kernel_fetch_bytes=pick(n, "expand_kernel")["fetch_bytes_x2"],
           expand_kernel_write_bytes=pick(n, "expand_kernel")["write_bytes"], reduce_kernel_hbm_bytes_per_launch=tot(pick(n, "reduce_kernel")),
           expand_sym_kernel_hbm_bytes_per_launch=tot(pick(s, "expand_sym_kernel")), expand_sym_kernel_write_bytes=pick(s, "expand_sym_kernel")["write_bytes"],
           rowsym_kernel_hbm_bytes_per_launch=tot(pick(s, "rowsym_kernel")), rowsym_kernel_write_bytes=pick(s, "rowsym_kernel")["write_bytes"],
           sym_product_hbm_bytes_total=own(s, ONE_SYM),
           mu16_expand_kernel_hbm_bytes_per_launch=tot(pick(m16, "expand_mfma16s")), mu16_reduce_kernel_hbm_bytes_per_launch=tot(pick(m16, "reduce_mfma16s")),
           mu16_product_hbm_bytes_total=own(m16, MU_N),
           transT_colsum_kernel_hbm_bytes_per_launch=tot(pick(tT, "expand_sym_kernel")), transT_rowsym_kernel_hbm_bytes_per_launch=tot(pick(tT, "rowsym_kernel")),
           transT_product_hbm_bytes_total=sum(tot(v) for k, v in tT.items() if any(s in k for s in ("expand_sym", "rowsym", "combine_list"))))
if s16:
    rec.update(sym_mu16_expand_kernel_hbm_bytes_per_launch=tot(pick(s16, "expand_sym_mfma16")), sym_mu16_rowsym_kernel_hbm_bytes_per_launch=tot(pick(s16, "rowsym_mfma16")),
               sym_mu16_product_hbm_bytes_total=own(s16, MU_SYM))
if c5:
    rec.update(c5_rank3_expand_kernel_hbm_bytes_per_launch=tot(pick(c5, "expand")), c5_rank3_product_hbm_bytes_total=own(c5, MU_SYM))
json.dump(rec, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
