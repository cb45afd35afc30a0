"""Summarise a rocprofv3 --pmc run: per kernel name, mean of each counter.  usage: pmc_summary.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "reduce_kernel" in k or "expand_kernel" in k or "read16" in k or "copy16" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-40s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
