#!/bin/bash
# final check of the round: smoke(), the full GPU suite, the drop-in calls in a fresh process, the c6 line with the attention kernel
out=$GRAFT_REPO_ROOT/gpurun_out/r04_c27; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
( time timeout 1500 python -m pytest tests -q -m gpu ) > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.txt; tail -6 $out/pytest_gpu.txt
timeout 300 python tools/api_rates.py c2 3 > $out/r04_api_path_standalone.json 2> $out/api.err; cat $out/r04_api_path_standalone.json
timeout 600 python bench.py --steps 20 --warmup 5 --preset c6 --no-cpu-baseline --no-reid-check > $out/r04_bench_c6_s1.json 2>$out/bench_c6.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04_c27/r04_bench_c6_s1.json').read().strip().splitlines()[-1])
print('c6', d['value'], d['ms_per_step'], d['frames_bit_exact'], d['net_outputs_check']['head_tensor_equal_to_eager_rerun'])
PY
