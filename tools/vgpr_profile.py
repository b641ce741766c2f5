"""Crude register-pressure profile of one kernel from `hipcc -S --cuda-device-only` output: for each window of
the ISA, the highest VGPR index referenced, with the labels / branches in the window.
usage: vgpr_profile.py file.s <substring of mangled kernel name> [window]"""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
win = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and key in l.split(":")[0] and ": ;" in l)
end = next(i for i in range(start, len(text)) if ".amdhsa_kernel" in text[i])
lines = text[start:end]
hi = []
for l in lines:
    regs = [int(x) for x in re.findall(r"\bv(\d+)\b", l)]
    regs += [int(b) for _, b in re.findall(r"v\[(\d+):(\d+)\]", l)]
    hi.append(max(regs) if regs else -1)
print(len(lines), "lines, max", max(hi))
for s in range(0, len(lines), win):
    labs = [l.strip().split()[0] for l in lines[s:s + win] if l.startswith(".LBB")]
    ops = {}
    for l in lines[s:s + win]:
        t = l.strip().split()
        if t and t[0].startswith(("v_", "ds_", "global_", "s_cbranch", "scratch_")):
            kk = t[0].split("_")[0] + "_" + t[0].split("_")[1]
            ops[kk] = ops.get(kk, 0) + 1
    top = " ".join(f"{k}:{v}" for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:5])
    print(f"{s:5d} {max(hi[s:s + win]):4d}  {' '.join(labs)[:40]:40s} {top}")
