#!/bin/bash
# cost attribution on the GPU box: repeat one part of cast_flat (MCRT_EXP build) and read the kernel time
cd $GRAFT_REPO_ROOT
for e in 1,1,1,1 2,1,1,1 1,2,1,1 1,1,2,1 1,1,1,2; do
  MCRT_LIB=$GRAFT_REPO_ROOT/build/libmcrt_exp.so MCRT_EXP=$e timeout -k 10 200 python bench.py --steps 2 --no-cpu-baseline --no-secondary --no-pmc-fields ${1:+--workload $1} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['roofline']['kernel_ms'], d['value'])" || exit 1
done
