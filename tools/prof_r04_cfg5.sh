# rocprofv3 passes of round 4 on BASELINE config 5 as worded (one GPU, every channel as pairs); run on the GPU box:
#   bash tools/prof_r04_cfg5.sh      (outputs under gpurun_out/prof_r04f/, summaries copied to profiles/ by hand)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/prof_r04f
mkdir -p $O
CMD="python3 bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-sides"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5_stats -- $CMD > $O/cfg5_stats.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/cfg5_sq1 -- $CMD > $O/cfg5_sq1.log 2>&1
timeout 900 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/cfg5_sq3 -- $CMD > $O/cfg5_sq3.log 2>&1
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/cfg5_sq5 -- $CMD > $O/cfg5_sq5.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cfg5_fetch -- $CMD > $O/cfg5_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cfg5_write -- $CMD > $O/cfg5_write.log 2>&1
for d in cfg5_sq1 cfg5_sq3 cfg5_sq5 cfg5_fetch cfg5_write; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 tools/pmc_sum.py $f > $O/$d.txt; done
f=$(find $O/cfg5_stats -name "*kernel_stats.csv" | head -1); cp $f $O/cfg5_kernel_stats.csv
grep -h "k_sweep1\|k_spline\|k_dynamics\|k_pointwise" $O/cfg5_*.txt
head -8 $O/cfg5_kernel_stats.csv
tail -1 $O/cfg5_stats.log | cut -c1-400
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
