#!/bin/bash
# MFMA utilisation of the conv path + association kernel: one PMC pass over a short bench run
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o run -- \
    python bench.py --graph none --overlap 0 --steps 20 --warmup 2 --no-cpu-baseline --no-batched --check-frames 0 > $out/bench.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
rows = []
for k, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0: continue
    # MFMA busy is summed over the 1024 SIMDs; utilisation = busy / (active cycles * 1024)
    rows.append((c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024), gui, cnt[k], k))
rows.sort(key=lambda r: -r[1])
print("mfma_util  gui_cycles_total  dispatches  kernel")
for u, g, n, k in rows[:25]:
    print(f"{u:8.4f}  {g:14.0f}  {n:6d}  {k}")
PY
find $out -name "*.csv" -size +6M -delete
