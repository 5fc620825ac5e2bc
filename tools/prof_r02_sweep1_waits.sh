cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r02
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/prof_r02/sq_counters.txt
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/prof_r02/w_$tag -- python3 tools/run_hotpath.py --workload ur6 --paths 1 --distinct 1 --reps 1 > gpurun_out/prof_r02/w_$tag.log 2>&1
  f=$(find gpurun_out/prof_r02/w_$tag -name "*counter_collection.csv" | head -1); python3 tools/pmc_sum.py $f | grep sweep
done
wc -l gpurun_out/prof_r02/sq_counters.txt
