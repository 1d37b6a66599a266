#!/bin/bash
# GPU box: SQ counter passes of one bench workload, summed per kernel family (all launches of one bench run)
# usage: tools/prof_kernels.sh <workload> <outtag>     (environment passes through)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --workload $1 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-pmc-fields"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/$2_a -- $B > /dev/null 2> $OUT/$2_a.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/$2_b -- $B > /dev/null 2> $OUT/$2_b.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$2_c -- $B > /dev/null 2> $OUT/$2_c.err
python3 - <<PY
import csv,glob,re
for p in "abc":
    for f in glob.glob("$OUT/$2_%s/*/*_counter_collection.csv"%p):
        per={}
        for r in csv.DictReader(open(f)):
            name=re.sub(r"\(.*","",r["Kernel_Name"].replace("void (anonymous namespace)::","").replace("(anonymous namespace)::",""))
            if not name.startswith("k_"): continue
            k=(name,r["Counter_Name"]); v=per.setdefault(k,[0.0,0.0,0,r["VGPR_Count"],r["Scratch_Size"],r["LDS_Block_Size"]])
            v[0]+=float(r["Counter_Value"]); v[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-6; v[2]+=1
        for (name,cn),v in sorted(per.items()):
            print("%-34s %-26s %14.6g  launches %5d  %9.2f ms  vgpr %s scratch %s lds %s"%(name,cn,v[0],v[2],v[1],v[3],v[4],v[5]))
PY
