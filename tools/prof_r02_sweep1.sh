cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ibatotp_amd/csrc -DBK_PROFILE_SECTIONS -shared -o /tmp/libdiag.so batotp_amd/csrc/batotp_hip.hip 2>&1 | grep error
for w in ${WORKLOADS:-ur6 gen7}; do echo "== $w"; python3 tools/sweep1_sections.py --lib /tmp/libdiag.so --workload $w --paths 1; done
