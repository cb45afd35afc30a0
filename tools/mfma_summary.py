"""Per kernel of a `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_<T> GRBM_GUI_ACTIVE` run: mean counters, mean
duration, and the fraction of the kernel's time its matrix cores were busy:
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles),  cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)
usage: mfma_summary.py <dir> <F64|F32>"""
import collections
import csv
import glob
import json
import sys

d, ty = sys.argv[1], sys.argv[2]
KEEP = ("mfma", "reduce", "expand", "rowsym", "combine")
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(s in k for s in KEEP):
            cnt[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(s in k for s in KEEP):
            dur[k].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
out = {}
for k, cs in sorted(cnt.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    rec = dict(m, launches=len(next(iter(cs.values()))))
    if dur.get(k):
        rec["dur_ns_under_pmc"] = sum(dur[k]) / len(dur[k])
    busy, gui = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), m.get("GRBM_GUI_ACTIVE", 0.0)
    if gui > 0:
        rec["cycles_per_xcd"] = gui / 8
        rec["mfma_busy_fraction"] = busy / (1024 * gui / 8)
    out[k] = rec
print(json.dumps(out, indent=1, sort_keys=True))
