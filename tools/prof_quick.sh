#!/bin/bash
# GPU box: two SQ counter passes of one workload with a chosen library build; prints per-kernel medians of k_render
# usage: tools/prof_quick.sh <tag-of-build|-> <workload> <outtag>
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out
[ "$1" != "-" ] && export MCRT_LIB=$ROOT/build/libmcrt_$1.so
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --workload $2 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-pmc-fields"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/$3_a -- $B > /dev/null 2> $OUT/$3_a.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/$3_b -- $B > /dev/null 2> $OUT/$3_b.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum --output-format csv -d $OUT/$3_c -- $B > /dev/null 2> $OUT/$3_c.err
python3 - <<PY
import csv,glob
for p in "abc":
    for f in glob.glob("$OUT/$3_%s/*/*_counter_collection.csv"%p):
        per={}
        for r in csv.DictReader(open(f)):
            if "k_render" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"],[]).append((float(r["Counter_Value"]),(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-6, r["VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"]))
        for k,v in per.items():
            top=max(x[0] for x in v); full=sorted(x for x in v if x[0]>0.5*top); m=full[len(full)//2]
            print("%-28s %14.6g   %8.2f ms vgpr %s scratch %s lds %s"%(k,m[0],m[1],m[2],m[3],m[4]))
PY
