#!/bin/bash
# per-kernel times of the small glue kernels before / after (ab_tmp/old.so = before)
cd $GRAFT_REPO_ROOT
SS_LIB_PATH=$PWD/ab_tmp/old.so bash tools/prof.sh old python tools/osnet_time.py 20 32 > /dev/null 2>&1
bash tools/prof.sh new python tools/osnet_time.py 20 32 > /dev/null 2>&1
python - <<'PY'
import csv,glob
for tag in ('old','new'):
    f=glob.glob(f'gpurun_out/prof_{tag}/*kernel_stats.csv')[0]
    r={x['Name'].split('(')[0]:x for x in csv.DictReader(open(f))}
    print(tag, {k: round(float(v['AverageNs'])/1e3,1) for k,v in r.items() if any(n in k for n in ('k_v8_decode','k_upcat','k_gate_vec','k_osnet_head','k_osnet_stem','k_sppf'))})
PY
