"""List VGPR / SGPR / scratch / LDS per kernel of libcmad_hip.so from the code-object metadata notes.

The library is linked from several translation units (cmad_amd/build.py), so its .hip_fatbin section holds one
offload bundle per unit; every bundle is unbundled and read.

    python tools/kernel_resources.py [libcmad_hip.so] [--summary]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so):
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so, os.path.join(tmp, "unused.o")], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, s in enumerate(starts):
        e = starts[i + 1] if i + 1 < len(starts) else len(blob)
        part = os.path.join(tmp, f"bundle{i}.bin")
        open(part, "wb").write(blob[s:e])
        co = os.path.join(tmp, f"dev{i}.co")
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        if os.path.getsize(co) > 0:
            out.append(co)
    return out


def kernels(so):
    rows = []
    for co in code_objects(so):
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        names = []
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)
            names.append((g("name").group(1), int(g("vgpr_count").group(1)), int(g("sgpr_count").group(1)),
                          int(g("private_segment_fixed_size").group(1)), int(g("group_segment_fixed_size").group(1)),
                          int(re.search(r"- \.agpr_count:\s+(\d+)", "- .agpr_count:" + blk).group(1))))
        dem = subprocess.run(["c++filt"], input="\n".join(n[0] for n in names), capture_output=True, text=True).stdout.split("\n")
        rows += [(d.strip(),) + n[1:] for d, n in zip(dem, names)]
    return rows


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    so = args[0] if args else "cmad_amd/csrc/libcmad_hip.so"
    rows = kernels(so)
    if "--summary" in sys.argv:
        fam = {}
        for r in rows:
            name = re.sub(r"^void \(anonymous namespace\)::", "", r[0]).split("<")[0].split("(")[0]
            f = fam.setdefault(name, [0, 0, 0, 0])
            f[0] += 1; f[1] = max(f[1], r[1]); f[2] += (r[3] > 0); f[3] = max(f[3], r[3])
        print(f"{len(rows)} kernels, {sum(1 for r in rows if r[3] > 0)} with scratch, library {os.path.getsize(so) / 1e6:.1f} MB")
        for k in sorted(fam):
            print(f"{fam[k][0]:5d} kernels  max {fam[k][1]:4d} vgpr  {fam[k][2]:4d} with scratch (max {fam[k][3]} B)  {k}")
    else:
        for r in sorted(rows):
            print(f"{r[1]:4d} vgpr {r[5]:4d} agpr {r[2]:4d} sgpr {r[3]:6d} scratch {r[4]:6d} lds  {r[0][:150]}")
