#!/bin/bash
# GPU box: parity subset, then config4_1024_16 for each value of MCRT_WALK_VOTE given (1024 = node + leaf step in every visit)
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py -m gpu -x -q --timeout 120 --timeout-method thread -k "octree or hits or config4 or fixture or more_objects" > gpurun_out/quick_oct_tests.log 2>&1 || { tail -20 gpurun_out/quick_oct_tests.log; exit 1; }
tail -1 gpurun_out/quick_oct_tests.log
for V in "$@"; do
  MCRT_WALK_VOTE=$V timeout -k 10 300 python bench.py --workload config4_1024_16 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/sweep_vote_$V.json 2> gpurun_out/sweep_vote_$V.err || { tail -5 gpurun_out/sweep_vote_$V.err; exit 1; }
  python - $V <<'PY'
import json,sys
d=json.loads(open('gpurun_out/sweep_vote_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('vote', sys.argv[1], d['ms_per_step'], 'ms', d['value'], 'Msamples/s')
PY
done
