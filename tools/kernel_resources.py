#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/scratch/occupancy per kernel of a .hip file (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                      "-ffp-contract=on", "-w", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for k, r in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{name:55s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>4} SGPR {r.get('TotalSGPRs','?'):>4} "
          f"scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?'):>2} LDS {r.get('LDS Size [bytes/block]','?')}")
