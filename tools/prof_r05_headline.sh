# rocprofv3 passes of round 5: the headline configuration (bench.py --config fill7 without side measurements) and the BASELINE
# configurations as worded; run on the GPU box:
#   bash tools/prof_r05h_headline.sh      (outputs under gpurun_out/prof_r05h/, summaries copied to profiles/ by hand)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r05o
mkdir -p $O
CMD="python3 bench.py --no-sides --no-as-worded --no-cpu-baseline --steps 2 --warmup 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fill7_stats -- $CMD > $O/fill7_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fill7_fetch -- $CMD > $O/fill7_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/fill7_write -- $CMD > $O/fill7_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/fill7_sq1 -- $CMD > $O/fill7_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/fill7_sq3 -- $CMD > $O/fill7_sq3.log 2>&1
for d in fill7_fetch fill7_write fill7_sq1 fill7_sq3; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 tools/pmc_sum.py $f > $O/$d.txt; done
f=$(find $O/fill7_stats -name "*kernel_stats.csv" | head -1); cp $f $O/fill7_kernel_stats.csv
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/fill7_sq4 -- $CMD > $O/fill7_sq4.log 2>&1
f=$(find $O/fill7_sq4 -name "*counter_collection.csv" | head -1); python3 tools/pmc_sum.py $f > $O/fill7_sq4.txt
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/fill7_sq5 -- $CMD > $O/fill7_sq5.log 2>&1
f=$(find $O/fill7_sq5 -name "*counter_collection.csv" | head -1); python3 tools/pmc_sum.py $f > $O/fill7_sq5.txt
grep -h "k_sweep\|k_spline" $O/fill7_*.txt
head -8 $O/fill7_kernel_stats.csv
# keep the merged output small
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
