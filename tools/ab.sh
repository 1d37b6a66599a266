#!/bin/bash
# GPU box: same-box A/B timing of library builds under build/ab/ on one workload.
#   usage: tools/ab.sh <workload> <steps> name[:ENV=VAL[,ENV=VAL]] ...     (each variant is run twice, interleaved)
cd $GRAFT_REPO_ROOT
W=$1; S=$2; shift 2
for round in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  out=gpurun_out/ab_${name}_$(echo "$envs" | tr ' =' '__')_$round
  env $envs MCRT_LIB=$GRAFT_REPO_ROOT/build/ab/$name.so timeout -k 10 300 python bench.py --workload $W --steps $S --warmup 1 --no-cpu-baseline --no-secondary --no-pmc-fields > $out.json 2> $out.err || { tail -5 $out.err; exit 1; }
  python - "$out.json" "$spec" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d['ms_per_step'], 'ms', d['value'])
PY
done
done
