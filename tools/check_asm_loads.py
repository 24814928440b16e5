#!/usr/bin/env python3
"""Static check of the fused kernel's inline-asm prefetch: between an asm `global_load` and the next
asm `s_waitcnt vmcnt`, no instruction may read or overwrite the load's destination registers (the
compiler does not know they are still in flight).  Usage: check_asm_loads.py <kernel.s>"""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
inflight = {}   # reg number -> line of the load
bad = 0
in_asm = False
for ln, text in enumerate(lines, 1):
    t = text.strip()
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if t.startswith("s_endpgm"):
        if inflight:
            print(f"line {ln}: kernel ends with loads never waited for"); bad += 1
        inflight.clear()
        continue
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    if in_asm and t.startswith("s_waitcnt vmcnt"):
        inflight.clear()
        continue
    regs = set()   # arch VGPR n -> n, accumulation VGPR n -> 1000 + n
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", t):
        base = 1000 if m.group(1) == "a" else 0
        regs.update(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    for m in re.finditer(r"\b([va])(\d+)\b", t):
        regs.add((1000 if m.group(1) == "a" else 0) + int(m.group(2)))
    if in_asm and t.startswith("global_load"):
        m = re.match(r"global_load_dword(?:x[234])? ([va])(?:\[(\d+):(\d+)\]|(\d+))", t)
        base = 1000 if m.group(1) == "a" else 0
        lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(2) else (int(m.group(4)), int(m.group(4)))
        dst = set(range(base + lo, base + hi + 1))
        for r in dst:
            if r in inflight:
                print(f"line {ln}: load overwrites in-flight v{r} (loaded at line {inflight[r]})"); bad += 1
        for r in dst:
            inflight[r] = ln
        continue
    hit = regs & set(inflight)
    if hit and not t.startswith("s_"):
        print(f"line {ln}: `{t}` touches in-flight {sorted(hit)} (loaded at {sorted(set(inflight[r] for r in hit))})")
        bad += 1
print("asm prefetch check:", "OK" if not bad else f"{bad} violations")
sys.exit(1 if bad else 0)
