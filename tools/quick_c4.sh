#!/bin/bash
# GPU box: the 1M-triangle bench line only (ms per step); optional env passes through
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --workload config4_1024_16 --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-pmc-fields > gpurun_out/quick_c4.json 2> gpurun_out/quick_c4.err || { tail -5 gpurun_out/quick_c4.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/quick_c4.json').read().strip().splitlines()[-1])
print('config4_1024_16', d['ms_per_step'], 'ms', d['value'], 'Msamples/s kernel_ms', d['roofline']['kernel_ms'])
PY
