# rocprofv3 passes of round 6 (final library): the headline configuration (bench.py --config fill7 without side measurements);
# run on the GPU box: bash tools/prof_r06_headline.sh   (outputs under gpurun_out/prof_r06/, summaries copied to profiles/ by hand)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r06
mkdir -p $O
CMD="python3 bench.py --no-sides --no-as-worded --no-cpu-baseline --steps 2 --warmup 1"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fill7_stats -- $CMD > $O/fill7_stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fill7_fetch -- $CMD > $O/fill7_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/fill7_write -- $CMD > $O/fill7_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/fill7_sq1 -- $CMD > $O/fill7_sq1.log 2>&1
timeout 900 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/fill7_sq3 -- $CMD > $O/fill7_sq3.log 2>&1
for d in fill7_fetch fill7_write fill7_sq1 fill7_sq3; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 tools/pmc_sum.py $f > $O/$d.txt; done
f=$(find $O/fill7_stats -name "*kernel_stats.csv" | head -1); cp $f $O/fill7_kernel_stats.csv
tail -1 $O/fill7_stats.log | cut -c1-600
grep -h "k_sweep\|k_spline\|k_pointwise" $O/fill7_*.txt
head -12 $O/fill7_kernel_stats.csv
# keep the merged output small
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
