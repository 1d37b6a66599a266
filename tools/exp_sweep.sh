#!/bin/bash
# GPU box: kernel time of several builds of the library (build/libmcrt_<tag>.so) on several workloads
#   usage: tools/exp_sweep.sh "<tags>" "<workloads>"
cd $GRAFT_REPO_ROOT
for w in $2; do
  for t in $1; do
    MCRT_LIB=$GRAFT_REPO_ROOT/build/libmcrt_$t.so timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-secondary --no-pmc-fields --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', '$t', d['roofline']['kernel_ms'], d['value'])" || exit 1
  done
done
