#!/bin/bash
# GPU box: kernel time of one library build under several environment settings ("A=1 B=2" per case, cases separated by ';')
cd $GRAFT_REPO_ROOT
IFS=';' read -ra CASES <<< "$3"
for c in "${CASES[@]}"; do
  LIBSET=""; [ "$1" != "-" ] && LIBSET="MCRT_LIB=$GRAFT_REPO_ROOT/build/libmcrt_$1.so"
  env $LIBSET $c timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-secondary --no-pmc-fields --workload $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2 $1 [$c]', d['roofline']['kernel_ms'], d['value'])" || exit 1
done
