#!/usr/bin/env python3
"""developer tool: VGPRs / spills / occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage)"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "extpom_amd", "csrc")
files = sys.argv[1:] or [f for f in sorted(os.listdir(SRC)) if f.endswith(".hip")]
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
                        "-I" + os.path.join(ROOT, "include"), "-I" + SRC, "-c", os.path.join(SRC, f), "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = None
    rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    for k, v in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
        print(f"{f:16s} {name:40s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} spill={v.get('VGPRs Spill')} sspill={v.get('SGPRs Spill')} scratch={v.get('ScratchSize')} occ={v.get('Occupancy')}")
