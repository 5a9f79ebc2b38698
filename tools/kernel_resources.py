#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage),
one line per kernel.  Usage: tools/kernel_resources.py sparch_amd/csrc/reccell.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
       "-Iinclude", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[bytes/\w+\])?: (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    if flt and flt not in name:
        continue
    print(f"{name:48s} vgpr {r.get('VGPRs', '?'):>4} agpr {r.get('AGPRs', '?'):>3} sgpr {r.get('SGPRs', '?'):>3} "
          f"scratch {r.get('ScratchSize', '?'):>4} lds {r.get('LDS Size', '?'):>6} occ {r.get('Occupancy', '?')}")
