#!/bin/bash
# Round-6 evidence, part B (run first): PMC of the association kernel (tracker workloads c2: 1 / 32 streams, c4: 1 stream), of the fp32
# ReID kernels and of the fp32 detector kernels; kernel stats of both fp32 networks.  Summaries -> gpurun_out/r06_prof/.
out=$GRAFT_REPO_ROOT/gpurun_out/r06_prof; mkdir -p $out; cd $GRAFT_REPO_ROOT
bash tools/pmc_assoc.sh c2_s1_f32 1 32 > $out/pmc_c2_s1_f32.txt 2>&1
bash tools/pmc_assoc.sh c2_b32_f32 32 32 > $out/pmc_c2_b32_f32.txt 2>&1
bash tools/pmc_assoc.sh c4_s1_f32 1 32 k_assoc 100 1920 1080 > $out/pmc_c4_s1_f32.txt 2>&1
python - <<'PY'
import json, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
res = {"_note": "rocprofv3 --pmc passes (tools/pmc_assoc.sh: SQ group, TCC hit/miss, FETCH_SIZE, WRITE_SIZE in separate runs) over tools/batched_assoc.py <streams> <frame_batch> [identities W H] (tracker-only loop, 160 frames: galleries full), round-6 final binary (k_assoc as in round 5; k_group_prep rotates the remainder tiles of a pair over its ranges); means over the last 3 k_assoc launches. hbm_bytes_per_launch = FETCH_SIZE*1024*2 (gfx950 counts 64 B per 128-B request, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024. Keys: <workload>_s<streams>_f<frames> (bench.py roofline.traffic) and <workload>_b<streams>_f<frames> (roofline_batched); workload c2 = 30 identities at 1280x720 (presets c2, c3, c5, c6 share it), c4 = 100 identities at 1920x1080."}
for tag in ("c2_s1_f32", "c2_b32_f32", "c4_s1_f32"):
    try:
        res[tag] = json.load(open(f"{root}/pmc_{tag}/summary.json"))
    except Exception as e:
        res[tag] = {"error": str(e)}
json.dump(res, open(root + "/r06_prof/r06_pmc_assoc.json", "w"), indent=1)
for k, v in res.items():
    if k != "_note": print(k, {a: v.get(a) for a in ("hbm_bytes_per_launch", "l2_hit_rate", "mfma_busy_cycles_per_simd_cycle")})
PY
PMC_GROUPS=0,1,3,4 bash tools/pmc_run.sh r06_osnet32 python tools/osnet32_eager.py 3 1024 > $out/pmc_osnet32.txt 2>&1
cp gpurun_out/pmc_r06_osnet32/summary.json $out/r06_pmc_osnet32.json
PMC_GROUPS=0,1,3,4 bash tools/pmc_run.sh r06_det32 python tools/det32_eager.py 3 32 > $out/pmc_det32.txt 2>&1
cp gpurun_out/pmc_r06_det32/summary.json $out/r06_pmc_det32.json
bash tools/prof.sh r06_osnet32 python tools/osnet32_time.py 5 1024 1 > $out/prof_osnet32.txt 2>&1
cp $(find gpurun_out/prof_r06_osnet32 -name "*kernel_stats.csv" | head -1) $out/r06_osnet32_kernel_stats.csv
grep crops gpurun_out/prof_r06_osnet32/cmd.log > $out/r06_osnet32_time.json
bash tools/prof.sh r06_det32 python tools/det32_eager.py 4 32 > $out/prof_det32.txt 2>&1
cp $(find gpurun_out/prof_r06_det32 -name "*kernel_stats.csv" | head -1) $out/r06_det32_kernel_stats.csv
( python tools/det32_eager.py 10 32; SS32_DET=0 python tools/det32_eager.py 10 32 ) 2>&1 | grep "^ok" > $out/r06_det32_time.txt
cat $out/r06_osnet32_time.json $out/r06_det32_time.txt
python - <<'PY'
import json, os
for f in ("r06_pmc_osnet32.json", "r06_pmc_det32.json"):
    d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06_prof/" + f))
    for k, e in sorted(d.items(), key=lambda kv: -(kv[1].get("pct_of_gpu_time") or 0))[:12]:
        print(k, e.get("avg_us"), e.get("mfma_busy_frac"), e.get("hbm_MB"), e.get("hbm_TBps"))
PY
