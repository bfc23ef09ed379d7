#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/prof.sh chain python tools/tracker_only.py > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_chain/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:50].ljust(52), r['Calls'], round(float(r['AverageNs'])/1e3,2), r['Percentage'])
t=glob.glob('gpurun_out/prof_chain/*kernel_trace.csv')
if t:
    rows=sorted(csv.DictReader(open(t[0])), key=lambda r:int(r['Start_Timestamp']))
    # gaps between consecutive chain kernels in the last 200 launches
    rows=[r for r in rows if r['Kernel_Name'].startswith(('k_frame','k_postnew'))][-400:]
    import statistics
    gaps=[(int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']))/1e3 for i in range(len(rows)-1)]
    gf=[g for g,r in zip(gaps,rows) if r['Kernel_Name'].startswith('k_frame')]
    gp=[g for g,r in zip(gaps,rows) if r['Kernel_Name'].startswith('k_postnew')]
    print('gap after k_frame median us', statistics.median(gf), 'after k_postnew', statistics.median(gp))
PY
