#!/bin/bash
# Round-6 evidence, part C (after the last change to the fp32 ReID kernels): PMC passes + kernel stats of the fp32 OSNet only.
out=$GRAFT_REPO_ROOT/gpurun_out/r06_prof; mkdir -p $out; cd $GRAFT_REPO_ROOT
PMC_GROUPS=0,1,3,4 bash tools/pmc_run.sh r06_osnet32 python tools/osnet32_eager.py 3 1024 > $out/pmc_osnet32.txt 2>&1
cp gpurun_out/pmc_r06_osnet32/summary.json $out/r06_pmc_osnet32.json
bash tools/prof.sh r06_osnet32 python tools/osnet32_time.py 5 1024 1 > $out/prof_osnet32.txt 2>&1
cp $(find gpurun_out/prof_r06_osnet32 -name "*kernel_stats.csv" | head -1) $out/r06_osnet32_kernel_stats.csv
grep crops gpurun_out/prof_r06_osnet32/cmd.log > $out/r06_osnet32_time.json
python tools/stem32_time.py 1024,860,430,28 2>&1 | grep -v amdgpu > $out/r06_stem32_time.txt
python tools/crop_time.py 2>&1 | grep -v amdgpu > $out/r06_crop_time.txt
cat $out/r06_osnet32_time.json $out/r06_stem32_time.txt
python - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06_prof/r06_pmc_osnet32.json"))
for k, e in sorted(d.items(), key=lambda kv: -(kv[1].get("pct_of_gpu_time") or 0))[:14]:
    print(k, e.get("avg_us"), e.get("mfma_busy_frac"), e.get("hbm_MB"), e.get("hbm_TBps"))
PY
