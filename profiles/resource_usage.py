#!/usr/bin/env python3
"""Parses hipcc -Rpass-analysis=kernel-resource-usage output (stderr of a compile) into one line per kernel.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> ru.txt; resource_usage.py ru.txt [substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+: (?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = list(rows)
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
for n, d in zip(names, dem):
    if want and not any(w in d for w in want):
        continue
    r = rows[n]
    print(f"{d[:70]:70s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', 0):3d} SGPR {r.get('TotalSGPRs', r.get('SGPRs', -1)):4d} scratch {r.get('ScratchSize', -1):5d} occ {r.get('Occupancy', -1):2d} LDS {r.get('LDS Size', -1):6d}")
