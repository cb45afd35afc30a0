"""Summarise a rocprofv3 --pmc run: per kernel name, mean of each counter.  usage: pmc_summary.py <dir> [--json]"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
KEEP = ("reduce", "expand", "combine", "rowsym", "read16", "copy16")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(s in k for s in KEEP):
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
if "--json" in sys.argv:
    print(json.dumps({k: {c: dict(mean=sum(v) / len(v), n=len(v)) for c, v in cs.items()} for k, cs in acc.items()}, indent=1, sort_keys=True))
else:
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("   %-40s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
