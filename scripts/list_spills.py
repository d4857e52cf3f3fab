"""Every kernel of the shipped libfsgpu.so whose metadata note reports spilled registers (vgpr / sgpr) or a private segment."""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "frankensearch_amd", "libfsgpu.so")
LL = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([LL + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = blob.find(magic)
    total = 0
    rows = []
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + 24)
        cur = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, cur)
            triple = blob[cur + 24:cur + 24 + tl]
            cur += 24 + tl
            if b"gfx950" not in triple or size == 0:
                continue
            co = os.path.join(td, "dev.co")
            open(co, "wb").write(blob[at + off:at + off + size])
            notes = subprocess.check_output([LL + "llvm-readelf", "--notes", co], stderr=subprocess.DEVNULL).decode(errors="replace")
            for m in re.finditer(r"- \.agpr_count:.*?(?=\n\s+- \.agpr_count:|\namdhsa\.target|\Z)", notes, re.S):
                item = m.group(0)
                g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", item).group(1))
                name = re.search(r"\.name:\s+(\S+)", item).group(1)
                total += 1
                if g("vgpr_spill_count") or g("sgpr_spill_count") or g("private_segment_fixed_size"):
                    rows.append((name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
        at = blob.find(magic, at + 1)
    dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    print(f"{total} kernels, {len(rows)} with spills or a private segment")
    for r, d in sorted(zip(rows, dem), key=lambda x: -x[0][2]):
        print(f"vgpr {r[1]:3d}  vgpr_spill {r[2]:4d}  sgpr_spill {r[3]:3d}  private {r[4]:5d}  {d[:150]}")
