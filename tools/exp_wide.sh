#!/bin/bash
# GPU box: parity subset + config-4 timing of library builds build/libmcrt_<tag>.so
cd $GRAFT_REPO_ROOT
for t in $1; do
  export MCRT_LIB=$GRAFT_REPO_ROOT/build/libmcrt_$t.so
  timeout -k 5 100 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 40 --timeout-method thread -v -k "octree or per_sample_radiance or hbm_resident or shape or million or all_accel" 2>&1 | tail -4; test ${PIPESTATUS[0]} -eq 0 || exit 1
  for w in ${2:-config4_1024_16}; do
    timeout -k 10 300 python bench.py --steps 2 --no-cpu-baseline --no-secondary --no-pmc-fields --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', '$t', d['roofline']['kernel_ms'], d['value'])" || exit 1
  done
done
