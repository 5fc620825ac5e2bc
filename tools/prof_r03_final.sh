# rocprofv3 kernel statistics of the BASELINE configurations as worded with the final library of round 3; run on the GPU box:
#   bash tools/prof_r03_final.sh      (outputs under gpurun_out/prof_r03f/, summaries copied to profiles/ by hand)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r03f
mkdir -p $O
for c in cfg2 cfg3 cfg4 cfg5; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_stats -- python3 bench.py --config $c --no-sides --no-cpu-baseline --steps 3 --warmup 1 > $O/${c}_stats.log 2>&1
  f=$(find $O/${c}_stats -name "*kernel_stats.csv" | head -1); cp $f $O/${c}_kernel_stats.csv
done
head -4 $O/cfg*_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
