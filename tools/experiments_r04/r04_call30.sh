#!/bin/bash
# last call of the round: the full GPU suite on the final tree, refreshed c1 / c6 lines
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c30; mkdir -p $out; cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -q -m gpu ) > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.txt; tail -6 $out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
for p in c1 c6; do timeout 600 python bench.py --steps 20 --warmup 5 --preset $p --no-cpu-baseline --no-reid-check > $out/r04_bench_${p}_s1.json 2>$out/bench_$p.err; done
for p in c1 c6; do python - $out/r04_bench_${p}_s1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['frames_bit_exact'], d['net_outputs_check']['head_tensor_equal_to_eager_rerun'])
PY
done
