#!/usr/bin/env python3
"""tools/kres.py <file.hip> [contract=off] [extra flags]: registers / LDS / occupancy of every kernel in one translation
unit of splat_renderer_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel."""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "splat_renderer_amd", "csrc")
contract = sys.argv[2] if len(sys.argv) > 2 else "off"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-I../../include", f"-ffp-contract={contract}", *sys.argv[3:],
       "-Rpass-analysis=kernel-resource-usage", "-c", sys.argv[1], "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.rsplit(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    g = lambda k: r.get(k, "?")
    print(f"{dem[:64]:64s} VGPR {g('VGPRs'):>4s} AGPR {g('AGPRs'):>3s} SGPR {g('TotalSGPRs'):>4s} scratch {g('ScratchSize [bytes/lane]'):>4s} "
          f"occ {g('Occupancy [waves/SIMD]'):>2s} LDS {g('LDS Size [bytes/block]'):>6s}")
