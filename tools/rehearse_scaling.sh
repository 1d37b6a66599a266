#!/bin/bash
# GPU box (ONE GPU): walk bench.py's N > 1 code path end to end with every rank on GPU 0 and gloo (MCRT_BENCH_REHEARSE=1; the numbers mean
# nothing, the path does): strong scaling for BASELINE configs 3 and 5 (the latter at a reduced pass count), weak for configs 2 and 4.
# usage: tools/rehearse_scaling.sh <tag>
TAG=$1; cd $GRAFT_REPO_ROOT
export MCRT_BENCH_REHEARSE=1
r() { name=$1; shift; timeout -k 10 500 python bench.py "$@" --no-cpu-baseline > gpurun_out/${TAG}_rehearse_$name.json 2> gpurun_out/${TAG}_rehearse_$name.log || { tail -5 gpurun_out/${TAG}_rehearse_$name.log; exit 1; }; tail -c 400 gpurun_out/${TAG}_rehearse_$name.json | head -c 10 > /dev/null; python - gpurun_out/${TAG}_rehearse_$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], 'n_gpus', d['n_gpus'], d['scaling'], d['value'], 'Msamples/s', d['config']['workload'], 'passes', d['config']['passes_per_step'], 'total_spp', d['config']['total_spp'])
PY
}
r gpus2_strong_config3 --gpus 2 --scaling strong --workload config3_1024_1024 --steps 2 --warmup 1
r gpus2_strong_4k_256spp --gpus 2 --scaling strong --workload cornell_4k_4096 --spp 256 --steps 1 --warmup 1
r gpus4_strong_4k_128spp --gpus 4 --scaling strong --workload cornell_4k_4096 --spp 128 --steps 1 --warmup 0
r gpus2_weak_config4 --gpus 2 --scaling weak --workload config4_1024_16 --steps 2 --warmup 1
r gpus2_weak_config2 --gpus 2 --scaling weak --workload config2_1024_256 --steps 2 --warmup 1
