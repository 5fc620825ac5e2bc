# round 2, the one-path-per-wavefront kernel k_sweep1 on single trajectories: section shares (diagnostic build), SQ counters, kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r02
bash tools/prof_r02_sweep1.sh > gpurun_out/prof_r02/sweep1_sections.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/prof_r02/s1_sq -- python3 tools/run_hotpath.py --workload ur6 --paths 1 --distinct 1 --reps 1 > gpurun_out/prof_r02/s1_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/prof_r02/s1_sq2 -- python3 tools/run_hotpath.py --workload ur6 --paths 1 --distinct 1 --reps 1 > gpurun_out/prof_r02/s1_sq2.log 2>&1
for d in s1_sq s1_sq2; do f=$(find gpurun_out/prof_r02/$d -name "*counter_collection.csv" | head -1); python3 tools/pmc_sum.py $f > gpurun_out/prof_r02/$d.txt; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02/s1_stats -- python3 bench.py --config cfg2 --no-cpu-baseline --no-sides > gpurun_out/prof_r02/s1_stats.log 2>&1
cp $(find gpurun_out/prof_r02/s1_stats -name "*kernel_stats.csv" | head -1) gpurun_out/prof_r02/s1_cfg2_kernel_stats.csv
cat gpurun_out/prof_r02/sweep1_sections.txt; grep -i sweep gpurun_out/prof_r02/s1_sq.txt gpurun_out/prof_r02/s1_sq2.txt; head -5 gpurun_out/prof_r02/s1_cfg2_kernel_stats.csv
