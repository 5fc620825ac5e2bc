# round 4, the one-path-per-wavefront kernel k_sweep1 on one KUKA LWR IV+ trajectory with torque limits (BASELINE config 3 as worded): SQ counters of the final
# library, the same two passes as round 2's profiles/r02_e_cfg2_sweep1_sq.txt part (c); run on the GPU box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r04s
mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/s1_sq -- python3 tools/run_hotpath.py --workload kuka7trq --paths 1 --distinct 1 --reps 1 > $O/s1_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/s1_sq2 -- python3 tools/run_hotpath.py --workload kuka7trq --paths 1 --distinct 1 --reps 1 > $O/s1_sq2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_IFETCH --kernel-trace --output-format csv -d $O/s1_sq3 -- python3 tools/run_hotpath.py --workload kuka7trq --paths 1 --distinct 1 --reps 1 > $O/s1_sq3.log 2>&1
for d in s1_sq s1_sq2 s1_sq3; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); python3 tools/pmc_sum.py $f > $O/$d.txt; done
grep -i sweep1 $O/s1_sq.txt $O/s1_sq2.txt $O/s1_sq3.txt
tail -3 $O/s1_sq.log
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
