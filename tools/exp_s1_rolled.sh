cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ibatotp_amd/csrc -DS1_ROLLED -shared -o /tmp/librolled.so batotp_amd/csrc/batotp_hip.hip 2>&1 | grep error
for w in ur6 gen7; do
echo "== $w unrolled"; python3 tools/run_hotpath.py --workload $w --paths 1 --distinct 1 --reps 2 --knots 50000 | tail -1
echo "== $w rolled"; python3 tools/run_hotpath.py --workload $w --paths 1 --distinct 1 --reps 2 --knots 50000 --lib /tmp/librolled.so | tail -1
done
