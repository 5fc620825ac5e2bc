set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r02
# 1) single trajectory (cfg2): instruction counters of the sweep kernel
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/prof_r02/cfg2_sq -- python3 tools/run_hotpath.py --workload ur6 --paths 1 --distinct 1 --reps 1 > gpurun_out/prof_r02/cfg2_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/prof_r02/cfg2_sq2 -- python3 tools/run_hotpath.py --workload ur6 --paths 1 --distinct 1 --reps 1 > gpurun_out/prof_r02/cfg2_sq2.log 2>&1
for d in cfg2_sq cfg2_sq2; do f=$(find gpurun_out/prof_r02/$d -name "*counter_collection.csv" | head -1); python3 tools/pmc_sum.py $f > gpurun_out/prof_r02/$d.txt; done
cat gpurun_out/prof_r02/cfg2_sq.txt gpurun_out/prof_r02/cfg2_sq2.txt | grep -i sweep
tail -3 gpurun_out/prof_r02/cfg2_sq.log
