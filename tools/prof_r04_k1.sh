# rocprofv3 passes of round 4 on the headline command for K1 (the single-pass spline kernel of large pair batches); run on the GPU box
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/prof_r04k
mkdir -p $O
CMD="python3 bench.py --no-sides --no-as-worded --no-cpu-baseline --steps 2 --warmup 1 --distinct 256"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1
for d in fetch write; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 tools/pmc_sum.py $f > $O/$d.txt; done
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv
grep -h "k_spline\|k_tile" $O/fetch.txt $O/write.txt
grep "k_spline\|k_tile\|k_sweep8\|k_pointwise" $O/kernel_stats.csv | cut -c1-160
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
