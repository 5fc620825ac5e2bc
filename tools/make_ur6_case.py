import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from batotp_amd import pathgen
w = bench.WORKLOADS["ur6"]
theta, cart, tres = w["gen"](1000, int(round(100000 / w["knots_per_coarse"])))
pathgen.write_traj_bin("path.dat", tres, theta, cart)
cfg = dict(w["cfg"]); cfg["out_smooth"] = 5
pathgen.write_config("config.dat", **cfg)
