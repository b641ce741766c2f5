"""Static instruction count of one kernel attributed to source lines / functions, from `hipcc -gline-tables-only -S
--cuda-device-only` output.  usage: isa_by_source.py file.s <kernel name substring> [topN]"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files = {}
for l in text:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m:
        files[int(m.group(1))] = m.group(2)
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and key in l.split(":")[0] and ": ;" in l)
end = next(i for i in range(start, len(text)) if ".amdhsa_kernel" in text[i])
cur = ("?", 0)
by_line = collections.Counter()
by_file = collections.Counter()
n = 0
for l in text[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = l.strip().split()
    if t and re.match(r"^(v_|s_|ds_|global_|buffer_|scratch_|flat_)", t[0]):
        by_line[cur] += 1
        by_file[cur[0]] += 1
        n += 1
print("instructions:", n)
for f, c in by_file.most_common():
    print(f"  {c:6d}  {f}")
print("top source lines:")
for (f, ln), c in by_line.most_common(top):
    print(f"  {c:5d}  {f}:{ln}")
