#!/bin/bash
# usage: tools/pmc_run.sh <tag> <command...>
# rocprofv3 --pmc passes (SQ activity, LDS, L2 hit, FETCH_SIZE, WRITE_SIZE) + a kernel-trace pass for durations over the
# same command; prints per-kernel means (kernels named k_*) and writes gpurun_out/pmc_<tag>/summary.json.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $root; mkdir -p $root
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  out=$root/p$i; mkdir -p $out
  if [ -n "$PMC_GROUPS" ] && [[ ",$PMC_GROUPS," != *",$i,"* ]]; then i=$((i+1)); continue; fi     # PMC_GROUPS=0,3,4: only those passes
  (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o run -- "$@" > $out/log.txt 2>&1)
  i=$((i+1))
done
out=$root/trace; mkdir -p $out
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- "$@" > $out/log.txt 2>&1)
python - "$root" <<'PY'
import csv, glob, json, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k"): continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0].replace("void ", "")
        if k.startswith("k"): dur[k] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["Percentage"]))
res = {}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    e = {"avg_us": dur.get(k, (None,))[0], "calls": dur.get(k, (0, 0))[1], "pct_of_gpu_time": dur.get(k, (0, 0, None))[2]}
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0                         # summed over the 8 XCDs
    if gui > 0:
        e["mfma_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024), 4)      # per-SIMD busy cycles / 1024 SIMDs
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_MB"] = round((m["FETCH_SIZE"] * 2 + m["WRITE_SIZE"]) * 1024 / 1e6, 2)            # FETCH_SIZE x2: gfx950 counts 64 of 128 B
        if e["avg_us"]: e["hbm_TBps"] = round(e["hbm_MB"] / e["avg_us"], 3)
    if "TCC_HIT_sum" in m: e["l2_hit"] = round(m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1), 3)
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"): e[n + "/wave_cycles"] = round(m.get(n, 0) / wc, 3)
    bc = m.get("SQ_BUSY_CYCLES", 0)
    for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VMEM"):
        if n in m: e[n] = round(m[n])
    res[k] = e
json.dump(res, open(root + "/summary.json", "w"), indent=1)
for k, e in sorted(res.items(), key=lambda kv: -(kv[1]["pct_of_gpu_time"] or 0))[:14]:
    print(k, json.dumps(e))
PY
find $root -name "*.csv" -size +4M -delete
