#!/usr/bin/env python3
"""dev: VGPRs / scratch / occupancy of every kernel of a .hip file (hipcc -S, device only): python tools/kernel_regs.py mg_table.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "modimizer_amd", "csrc")
out = "/tmp/%s.s" % os.path.basename(src)
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(csrc, "..", "..", "include"), "-I" + csrc,
                "--cuda-device-only", "-S", os.path.join(csrc, src), "-o", out] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
name = None
for line in open(out):
    m = re.match(r"\s*\.amdhsa_kernel (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        info = {}
    m = re.match(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", line)
    if m and name:
        info[m.group(1)] = int(m.group(2))
        if m.group(1) == "LDSByteSize" and flt in name:
            print("%-90s vgpr %3d agpr %3d scratch %4d occupancy %d" % (name[:90], info.get("NumVgprs", -1), info.get("NumAgprs", 0), info.get("ScratchSize", -1), info.get("Occupancy", -1)))
