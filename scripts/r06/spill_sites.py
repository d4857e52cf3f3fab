#!/usr/bin/env python3
"""Where a kernel's register spills sit: scratch_load / scratch_store instructions of the gfx950 ISA of a .hip file, per kernel, classified
by whether they lie inside a loop body (a backward branch's span) — a spill parked across a kernel's phases costs a store and a load per
thread and launch, one inside a k-loop costs them per iteration.
  usage: spill_sites.py <file.hip> [extra hipcc flags ...]   (compiles with the library's flags to ISA in a temp dir)"""
import os
import re
import subprocess
import sys
import tempfile

src = os.path.abspath(sys.argv[1])
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-x", "hip",
                           "-I", os.path.join(root, "include"), "-S", "--cuda-device-only", src, "-o", out] + sys.argv[2:], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for si, start in enumerate(starts):
    end = next((i for i in range(start, len(lines)) if "s_endpgm" in lines[i]), len(lines))
    if si + 1 < len(starts) and starts[si + 1] < end:
        continue
    body = lines[start:end]
    sc = [i for i, l in enumerate(body) if "scratch_" in l]
    if not sc:
        continue
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB[0-9_]+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB[0-9_]+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    inner = []
    for a, b in loops:   # innermost loops: those that contain no other loop
        if not any((c > a or d < b) and c >= a and d <= b for c, d in loops if (c, d) != (a, b)):
            inner.append((a, b))
    n_loop = sum(1 for i in sc if any(a <= i <= b for a, b in loops))
    n_inner = sum(1 for i in sc if any(a <= i <= b for a, b in inner))
    name = subprocess.run(["c++filt", lines[start].split(":")[0]], capture_output=True, text=True).stdout.strip()[:110]
    print(f"{name}\n    scratch instructions {len(sc)} ({sum('store' in body[i] for i in sc)} stores, {sum('load' in body[i] for i in sc)} loads); "
          f"inside any loop {n_loop}; inside an INNERMOST loop {n_inner}; v_mfma in the kernel {sum('v_mfma' in l for l in body)}")
