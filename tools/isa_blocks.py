#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (tools/isa_blocks.py file.s kernel-substring)."""
import re, sys
text = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l)
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
blocks, cur = [], ["entry", 0, 0, 0, 0]
blocks.append(cur)
for l in text[start + 1:end + 1]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), 0, 0, 0, 0]; blocks.append(cur); continue
    t = l.strip()
    if not t or t[0] in ";.":
        continue
    op = t.split()[0]
    cur[1 if op.startswith("v_") else 2 if op.startswith("s_") else 3 if op.startswith("ds_") else 4] += 1
for b in blocks:
    if b[1] + b[2] >= int(sys.argv[3]) if len(sys.argv) > 3 else 12:
        print("%-12s valu %3d salu %3d lds %2d mem %2d" % tuple(b))
print("total valu", sum(b[1] for b in blocks), "salu", sum(b[2] for b in blocks))
