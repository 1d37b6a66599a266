#!/bin/bash
# GPU box: the round's bench lines, one JSON file per workload under gpurun_out/<tag>_bench_<workload>.json
# usage: tools/final_bench.sh <tag>
TAG=$1; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || exit 1
for W in cornell_512_64 config3_1024_1024 cornell_4k_64 config4_1024_16; do
  timeout -k 10 400 python bench.py --workload $W --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_$W.json 2> gpurun_out/${TAG}_bench_$W.err || exit 1
done
timeout -k 10 900 python bench.py --workload config4_2048_256 --steps 2 > gpurun_out/${TAG}_bench_config4_2048_256.json 2> gpurun_out/${TAG}_bench_config4_2048_256.err || exit 1
for A in scan brute octree; do
  timeout -k 10 400 python bench.py --accel $A --steps 2 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_config2_$A.json 2> gpurun_out/${TAG}_bench_config2_$A.err || exit 1
done
MCRT_BENCH_REHEARSE=1 timeout -k 10 400 python bench.py --gpus 2 --steps 2 --no-cpu-baseline --no-secondary --workload cornell_512_64 > gpurun_out/${TAG}_bench_rehearse_gpus2.json 2> gpurun_out/${TAG}_bench_rehearse_gpus2.err || exit 1
MCRT_BENCH_REHEARSE=1 timeout -k 10 400 python bench.py --gpus 2 --steps 2 --no-cpu-baseline --no-secondary --workload config4_1024_16 > gpurun_out/${TAG}_bench_rehearse_gpus2_config4.json 2> gpurun_out/${TAG}_bench_rehearse_gpus2_config4.err || exit 1
for f in gpurun_out/${TAG}_bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['value'], d['n_gpus'], d['roofline']['bound'], d['roofline'].get('frac'), d['roofline']['kernel_ms'], (d.get('cpu_baseline') or {}).get('value'))
PY
done
