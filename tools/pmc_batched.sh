#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_cosine_stream at 32 streams per launch
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_b32_$ctr; rm -rf $out; mkdir -p $out
  (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python tools/batched_assoc.py 32 > $out/log.txt 2>&1)
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" "$ctr" <<'PY'
import csv, sys
f, ctr = sys.argv[1], sys.argv[2]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr and r["Kernel_Name"].startswith("k_cosine_stream")]
print(ctr, "k_cosine_stream dispatches", len(v), "last20 mean KiB", sum(v[-20:]) / max(len(v[-20:]), 1))
PY
  find $out -name "*.csv" -size +6M -delete
done
