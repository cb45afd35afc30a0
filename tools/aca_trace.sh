#!/bin/bash
# Runs ON THE GPU BOX: start / end / duration of every compression kernel of one bench.py build (rocprofv3 --kernel-trace); extra args go to bench.py
export TMPDIR=/tmp
cd /root/repo
rm -rf /tmp/aca_tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/aca_tr -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $* > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
rows=[]
for f in glob.glob('/tmp/aca_tr/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'aca' in r['Kernel_Name'] and 'aca_cb' not in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:], r.get('Grid_Size_X', r.get('Grid_Size',''))))
rows.sort()
t0=rows[0][0]
for a,b,n,g in rows[:60]:
    print('%9.3f -> %9.3f ms  (%8.3f)  %s grid %s'%((a-t0)/1e6,(b-t0)/1e6,(b-a)/1e6,n,g))
PY
