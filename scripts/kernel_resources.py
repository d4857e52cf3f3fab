#!/usr/bin/env python3
"""Registers / scratch / occupancy per kernel instantiation, from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(stdin = the compiler's stderr).  Usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2>&1 >/dev/null | kernel_resources.py"""
import re
import sys

cur, rows = None, {}
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+) \[", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for name, v in rows.items():
    targs = re.findall(r"L[ib](\d+)E", name)
    print(f"{name[:40]:40s} <{','.join(targs)}>  vgpr {v.get('VGPRs')}  agpr {v.get('AGPRs')}  spill {v.get('VGPRs Spill')}  "
          f"scratch {v.get('ScratchSize [bytes/lane]')}  occ {v.get('Occupancy [waves/SIMD]')}")
