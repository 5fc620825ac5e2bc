#!/bin/bash
# round 4: SQ counters of the batch sweep kernels on a reduced headline batch (16 384 GEN7DOF paths x 2e4 knots, one resident
# batch): k_sweep<8,-1,true,true> / k_sweep<8,-1,true,false> (round 3's reverse / forward) against k_sweep8<8,-1,-1> / <8,-1,1>
#   bash tools/prof_r04_sweep8.sh [variants]      (run on the GPU box; summaries under gpurun_out/prof_r04/)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r04
mkdir -p $O
V=${1:-0:4:-1:8:8,1:4:8:8:8}
CMD="python3 tools/run_hotpath.py --workload gen7 --paths 16384 --knots 20000 --distinct 64 --group 8 --reps 1 --variants $V"
pass() { # name, counters...
  n=$1; shift
  timeout 900 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- $CMD > $O/$n.log 2>&1
  f=$(find $O/$n -name "*counter_collection.csv" | head -1); python3 tools/pmc_sum.py $f | grep "k_sweep" > $O/$n.txt
  find $O/$n -name "*.csv" -delete; find $O/$n -name "*.db" -delete
}
pass sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU
pass sq3 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass sq4 SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv
find $O/stats -name "*.csv" -delete; find $O/stats -name "*.db" -delete
cat $O/*.txt; head -12 $O/kernel_stats.csv; tail -3 $O/stats.log
