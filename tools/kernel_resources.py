#!/usr/bin/env python3
"""Registers, scratch and LDS of the kernels in a built libbatotp_hip.so (from the code object's metadata notes).

    python tools/kernel_resources.py [batotp_amd/csrc/libbatotp_hip.so] [substring of the kernel name]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "batotp_amd", "csrc", "libbatotp_hip.so")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    syms = subprocess.check_output([f"{LLVM}/llvm-readelf", "--symbols", "--wide", co], text=True)
demangle = lambda n: subprocess.check_output(["c++filt", n], text=True).strip()
code = {}
for line in syms.splitlines():
    f = line.split()
    if len(f) == 8 and f[3] == "FUNC":
        code[f[7]] = int(f[2], 0)
rows = []
for blk in notes.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if pat and pat not in name and pat not in demangle(name):
        continue
    rows.append((demangle(name)[:110], g("vgpr_count"), blk.split("\n")[0].strip(), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"),
                 g("group_segment_fixed_size"), code.get(name, 0)))
print(f"{'kernel':110s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'spill':>5s} {'scratch':>7s} {'lds':>7s} {'code B':>8s}")
for r in sorted(rows):
    print(f"{r[0]:110s} {r[1]:>5s} {r[2]:>5s} {r[3]:>5s} {r[4]:>5s} {r[5]:>7s} {r[6]:>7s} {r[7]:8d}")
