#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 passes of `bench.py --workload W` whose summaries
# tools/summarize_profiles.py turns into profiles/<round>_*.  One pass per rocprofv3 run (counter slots:
# 8 SQ, 4 TCC - FETCH_SIZE takes 3, WRITE_SIZE 2 - per run), kernel-trace stats in a run of their own.
#   usage: tools/profile_pmc.sh <tag> <workload> [<workload> ...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY"
for W in "$@"; do
  B="python3 $ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-pmc-fields"
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_$W -- $B > $OUT/${TAG}_stats_$W.json 2> $OUT/${TAG}_stats_$W.err || exit 1
  timeout -s KILL 300 rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/${TAG}_sq1_$W -- $B > /dev/null 2> $OUT/${TAG}_sq1_$W.err || exit 1
  timeout -s KILL 300 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/${TAG}_sq2_$W -- $B > /dev/null 2> $OUT/${TAG}_sq2_$W.err || exit 1
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch_$W -- $B > /dev/null 2> $OUT/${TAG}_fetch_$W.err || exit 1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write_$W -- $B > /dev/null 2> $OUT/${TAG}_write_$W.err || exit 1
  timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_grbm_$W -- $B > /dev/null 2> $OUT/${TAG}_grbm_$W.err || exit 1
  echo "profiled $W"
done
