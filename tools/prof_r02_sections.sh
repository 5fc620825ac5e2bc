cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ibatotp_amd/csrc -DBK_PROFILE_SECTIONS -shared -o /tmp/libdiag.so batotp_amd/csrc/batotp_hip.hip 2>&1 | grep -E "error" | head
for w in ur6 gen7; do for p in 1 128; do echo "== $w paths $p"; python3 tools/sweep_sections.py --lib /tmp/libdiag.so --workload $w --paths $p --distinct 1 --knots 50000; done; done
