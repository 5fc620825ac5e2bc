# round 2: kernel statistics of the BASELINE configs 3, 4 and 5 as worded (one GPU); run on the GPU box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r02
mkdir -p $O
for c in cfg3 cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${c}_stats -- python3 bench.py --config $c --no-cpu-baseline --no-sides > $O/${c}_stats.log 2>&1
  cp $(find $O/${c}_stats -name "*kernel_stats.csv" | head -1) $O/${c}_kernel_stats.csv
  head -6 $O/${c}_kernel_stats.csv | cut -c1-200
  tail -1 $O/${c}_stats.log | cut -c1-300
done
find $O -name "*kernel_trace.csv" -delete
