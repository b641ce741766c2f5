"""List VGPR / SGPR / scratch / LDS per kernel of libcmad_hip.so from the code-object metadata notes."""
import re
import subprocess
import sys

so = sys.argv[1] if len(sys.argv) > 1 else "cmad_amd/csrc/libcmad_hip.so"
import os
import tempfile
tmp = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", f"--input={so}",
                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp}/dev.co"], check=False)
co = f"{tmp}/dev.co"
if not os.path.exists(co) or os.path.getsize(co) == 0:
    # fall back: extract .hip_fatbin section
    subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", f".hip_fatbin={tmp}/fat.bin", so], check=True)
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", f"--input={tmp}/fat.bin",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
rows = []
for blk in notes.split("- .agpr_count:")[1:]:
    g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)
    name = g("name").group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    rows.append((dem, int(g("vgpr_count").group(1)), int(g("sgpr_count").group(1)),
                 int(g("private_segment_fixed_size").group(1)), int(g("group_segment_fixed_size").group(1))))
for r in sorted(rows):
    print(f"{r[1]:4d} vgpr {r[2]:4d} sgpr {r[3]:6d} scratch {r[4]:6d} lds  {r[0][:110]}")
