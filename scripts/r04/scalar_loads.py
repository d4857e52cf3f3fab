#!/usr/bin/env python3
"""Every scalar memory load of libfsgpu.so's device code whose base is NOT the kernarg segment.

r03 verdict, item 4b: a scalar load (s_load_* / s_buffer_load_*) goes through the scalar data cache; the compiler selects one
for any wave-uniform address it can prove is not written by the same kernel, not only for kernel arguments.  This script
unbundles the gfx950 code objects of the built library, disassembles them, and lists per kernel the scalar loads whose base
register pair does not derive from the kernarg pointer, with the instructions that defined the base (so that the buffer can be
named).  Output: profiles/r04/scalar_loads.txt (the annotated copy adds who writes each buffer).

    python scripts/r04/scalar_loads.py [path/to/libfsgpu.so] > profiles/r04/scalar_loads_raw.txt
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SREG = re.compile(r"s\[(\d+):(\d+)\]|s(\d+)")


def sregs(tok: str):
    m = SREG.fullmatch(tok.strip().rstrip(","))
    if not m:
        return None
    if m.group(3) is not None:
        return (int(m.group(3)), int(m.group(3)))
    return (int(m.group(1)), int(m.group(2)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def analyse(fn_name: str, insns: list[str]):
    """returns [(index, text, base pair, provenance lines)] for the non-kernarg scalar loads of one function"""
    kern = None           # SGPR pair holding the kernarg pointer
    derived = set()       # pairs known to be kernarg + constant
    found = []
    defs: dict[int, int] = {}   # sgpr -> index of its last defining instruction
    for i, line in enumerate(insns):
        parts = line.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if op.startswith("s_load_") or op.startswith("s_buffer_load_"):
            base = sregs(ops[1])
            if kern is None and op.startswith("s_load_"):
                kern = base
                derived.add(base)
            if base not in derived:
                prov = []
                seen = set()
                work = list(range(base[0], base[1] + 1))
                depth = 0
                while work and depth < 12:
                    r = work.pop()
                    j = defs.get(r)
                    if j is None or j in seen:
                        continue
                    seen.add(j)
                    depth += 1
                    prov.append((j, insns[j]))
                    # follow scalar sources one level (to reach the kernarg load that produced the pointer)
                    src = insns[j].split(None, 1)
                    if len(src) > 1:
                        for tok in src[1].split(",")[1:]:
                            rr = sregs(tok)
                            if rr and rr != kern:
                                work.extend(range(rr[0], rr[1] + 1))
                found.append((i, line, base, sorted(prov)))
        # record scalar definitions (destination = first operand of s_* / v_readfirstlane / v_cmp..e64 etc.)
        if ops:
            d = sregs(ops[0])
            if d and (op.startswith("s_") or op.startswith("v_readfirstlane") or op.startswith("v_readlane")) and not op.startswith(
                    ("s_waitcnt", "s_cbranch", "s_branch", "s_barrier", "s_nop", "s_cmp", "s_bitcmp", "s_setprio", "s_sleep", "s_endpgm",
                     "s_dcache", "s_store", "s_setreg", "s_sendmsg", "s_trap", "s_icache")):
                for r in range(d[0], d[1] + 1):
                    defs[r] = i
                # kernarg + constant stays kernarg (the implicit-argument block)
                if op in ("s_add_u32", "s_mov_b64", "s_add_u64") and len(ops) >= 2:
                    s0 = sregs(ops[1])
                    if op == "s_mov_b64" and s0 in derived:
                        derived.add(d)
                    elif op == "s_add_u32" and kern and s0 == (kern[0], kern[0]) and i + 1 < len(insns):
                        nxt = insns[i + 1].split(None, 1)
                        if nxt[0] == "s_addc_u32":
                            nops = [o.strip() for o in nxt[1].split(",")]
                            if sregs(nops[1]) == (kern[1], kern[1]) and nops[2] == "0":
                                derived.add((d[0], sregs(nops[0])[0]))
                    else:
                        derived.discard(d)
                elif d in derived and d != kern:
                    derived.discard(d)
    return kern, found


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "frankensearch_amd", "libfsgpu.so")
    tmp = tempfile.mkdtemp(prefix="fsgpu_dis_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, capture_output=True, check=True)
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        total_kernels = total_loads = total_flagged = 0
        print(f"# scalar loads with a base other than the kernarg pointer, per kernel ({os.path.basename(lib)}, {len(objs)} gfx950 code objects)")
        print("# columns: instruction | base pair | the scalar instructions that defined the base (nearest first level of provenance)")
        for obj in objs:
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], cwd=tmp, capture_output=True,
                                 text=True, check=True).stdout
            funcs: list[tuple[str, list[str]]] = []
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    funcs.append((m.group(1), []))
                elif funcs and line.startswith("\t"):
                    text = line.split("//")[0].strip()
                    if text:
                        funcs[-1][1].append(text)
            names = demangle([f[0] for f in funcs])
            for name, insns in funcs:
                nload = sum(1 for l in insns if l.startswith(("s_load_", "s_buffer_load_")))
                if not nload:
                    continue
                kern, found = analyse(name, insns)
                total_kernels += 1
                total_loads += nload
                if not found:
                    continue
                total_flagged += len(found)
                print(f"\n## {names[name]}\n   kernarg pointer s[{kern[0]}:{kern[1]}], {nload} scalar loads, {len(found)} not from the kernarg segment")
                for i, line, base, prov in found:
                    print(f"   [{i:5d}] {line}")
                    for j, p in prov:
                        print(f"           <- [{j:5d}] {p}")
        print(f"\n# total: {total_kernels} kernels with scalar loads, {total_loads} scalar loads, {total_flagged} with a non-kernarg base")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
