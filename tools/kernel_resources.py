#!/usr/bin/env python3
"""VGPR / scratch / LDS / occupancy per kernel of one HIP source
(hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py vo_slam_test_amd/csrc/ba.hip [extra hipcc flags]"""
import re, subprocess, sys
src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", *extra,
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-40s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "occ"))
for k, v in rows.items():
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", k)[:40]
    print("%-40s %5d %5d %7d %6d %4d" % (name, v.get("VGPRs", 0), v.get("TotalSGPRs", 0),
          v.get("ScratchSize [bytes/lane]", 0), v.get("LDS Size [bytes/block]", 0), v.get("Occupancy [waves/SIMD]", 0)))
