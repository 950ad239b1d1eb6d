#!/bin/bash
# PMC look at one configuration under an environment: tools/qprof.sh <tag> <config> <frames_per_launch> "<ENV=..>"
TAG=$1; CFG=$2; FPL=${3:-4}; ENVS=${4:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CHILD="env $ENVS python $R/bench.py --pmc-child --config $CFG --steps 2 --warmup 1 --frames-per-launch $FPL"
rocprofv3 --list-avail 2>/dev/null | grep -o -i "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*\|SQ_[A-Z_]*FETCH[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/avail.txt
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_IFETCH SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$i -o pmc -- $CHILD > /dev/null 2> $OUT/pmc_$i.err
done
python - <<PY > $OUT/summary.txt 2>&1
import glob, sqlite3
print(open("$OUT/avail.txt").read())
for d in sorted(glob.glob("$OUT/pmc_*/**/*.db", recursive=True)):
    cur = sqlite3.connect(d).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(v) from (select kernel_name, counter_name, dispatch_id, sum(value) as v "
         "from counters_collection where kernel_name like '%rt_trace%' group by kernel_name, counter_name, dispatch_id) group by kernel_name, counter_name")
    for k, name, n, avg in cur.execute(q):
        print("%-40s %-26s n=%3d avg=%.6g" % (k[:40], name, n, avg))
    for k, n, avg in cur.execute("select name, count(*), avg(duration) from kernels where name like '%rt_trace%' group by name"):
        print("   dispatch %-40s n=%d avg_ms=%.3f" % (k[:40], n, avg / 1e6))
PY
rm -rf $OUT/pmc_*/
cat $OUT/summary.txt
