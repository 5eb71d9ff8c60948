#!/usr/bin/env python3
"""Reads hipcc's -Rpass-analysis=kernel-resource-usage remarks (stderr of a compile) and prints one line per kernel:
registers, spills, scratch, LDS, occupancy.  usage: kernel_usage.py remarks.txt"""
import re
import subprocess
import sys

cur = None
rows = []
for line in open(sys.argv[1]):
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:") or body.startswith("Name:"):
        name = body.split(":", 1)[1].strip()
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        cur = {"name": name}
        rows.append(cur)
    elif cur is not None and ":" in body:
        k, v = body.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    short = re.sub(r"\(cb::DrawArgs\)|cb::\(anonymous namespace\)::|void ", "", r["name"])
    print(f"{short:60s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('TotalSGPRs', r.get('SGPRs','?')):>4} "
          f"spill s/v {r.get('SGPRs Spill','?')}/{r.get('VGPRs Spill','?')} scratch {r.get('ScratchSize [bytes/lane]','?')} "
          f"LDS {r.get('LDS Size [bytes/block]','?')} occ {r.get('Occupancy [waves/SIMD]','?')}")
