#!/usr/bin/env python3
"""Instruction statistics of the loops of one kernel in a gfx950 .s file: VGPRs, scratch, and per inner loop the
instruction count, VALU count and the commonest mnemonics.   usage: isa_loop_stats.py <file.s> <substring of the kernel symbol>"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
want = sys.argv[2]
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and want in l]
for start in starts:
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    print(lines[start][:150])
    for l in lines[end:end + 60]:
        if re.search(r"; (NumVgprs|ScratchSize|Occupancy|NumSgprs):", l):
            print("  ", l.strip())
    body = lines[start:end]
    for i, l in enumerate(body):
        if "Loop Header" not in l:
            continue
        lab = l.split(":")[0]
        back = [k for k in range(i, len(body)) if re.search(r"s_cbranch\w+ " + re.escape(lab) + r"$", body[k].strip())]
        if not back:
            continue
        seg = [x.strip() for x in body[i + 1:back[-1] + 1] if x.strip() and not x.strip().startswith(";")]
        c = collections.Counter(x.split()[0] for x in seg)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print(f"  loop {lab}: {len(seg)} instructions, {valu} VALU")
        if len(seg) > 300:
            print("    " + ", ".join(f"{k} {v}" for k, v in c.most_common(24)))
