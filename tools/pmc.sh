#!/bin/bash
# usage: tools/pmc.sh <tag> <counter> <bench args...>  -> gpurun_out/pmc_<tag>_<counter>/  (counter_collection csv)
tag=$1; ctr=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_${ctr}
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python bench.py "$@" > $out/bench.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python - "$f" "$ctr" <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != ctr: continue
    k = r["Kernel_Name"].split("(")[0]
    if not k.startswith(("k_", "void k_")): continue
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(f"{ctr} {k}: dispatches {n} mean {v/n:.1f}")
PY
find $out -name "*.csv" -size +8M -delete
