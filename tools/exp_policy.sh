#!/bin/bash
# GPU box: LanePolicy sweep (MCRT_MIN_TRACE x MCRT_FETCH_MIN, runtime knobs) on one workload
cd $GRAFT_REPO_ROOT
for mt in $2; do for fm in $3; do
  MCRT_MIN_TRACE=$mt MCRT_FETCH_MIN=$fm timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-secondary --no-pmc-fields --workload $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 min_trace=$mt fetch_min=$fm', d['roofline']['kernel_ms'], d['value'])" || exit 1
done; done
