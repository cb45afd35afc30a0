"""Durations of the aca_team_* launches from a rocprofv3 --kernel-trace csv: every 64th iteration, and totals per kernel."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "aca" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("<")[0].split("::")[-1], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
tot = {}
for s, e, k, g in rows:
    tot.setdefault(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += (e - s) / 1e6
for k, (n, ms) in tot.items():
    print("%-28s %6d launches %9.1f ms" % (k, n, ms))
team = [r for r in rows if "team" in r[2]]
if team:
    print("team phase: first start -> last end %.1f ms, sum of kernel time %.1f ms" % ((team[-1][1] - team[0][0]) / 1e6, sum(e - s for s, e, _, _ in team) / 1e6))
    for i in range(0, len(team) - 2, 3 * 64):
        a = team[i:i + 3]
        print("launch %5d: " % i + "  ".join("%s %.3f ms (grid %d)" % (k.replace("aca_team_", "").replace("_kernel", ""), (e - s) / 1e6, g) for s, e, k, g in a) + "   gap to next %.3f ms" % ((team[i + 3][0] - a[2][1]) / 1e6 if i + 3 < len(team) else 0))
