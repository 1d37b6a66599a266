#!/bin/bash
# GPU box: the octree parity subset, then the 1M-triangle bench line (ms per step)
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py -m gpu -x -q --timeout 120 --timeout-method thread -k "octree or hits or config4 or fixture or more_objects" > gpurun_out/quick_oct_tests.log 2>&1 || { tail -20 gpurun_out/quick_oct_tests.log; exit 1; }
tail -1 gpurun_out/quick_oct_tests.log
timeout -k 10 300 python bench.py --workload config4_1024_16 --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/quick_oct_bench.json 2> gpurun_out/quick_oct_bench.err || { tail -5 gpurun_out/quick_oct_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/quick_oct_bench.json').read().strip().splitlines()[-1])
print('config4_1024_16', d['ms_per_step'], 'ms', d['value'], 'Msamples/s kernel_ms', d['roofline']['kernel_ms'])
PY
