#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of one bench workload (environment passes through); prints the per-kernel summary
# usage: tools/prof_stats.sh <workload> <outtag>
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$2 -- python3 $ROOT/bench.py --workload $1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-pmc-fields > $OUT/$2.json 2> $OUT/$2.err || exit 1
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/$2/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-90s calls %7s total %12.3f ms avg %10.1f us  %5s%%"%(r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
