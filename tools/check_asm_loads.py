#!/usr/bin/env python3
"""Static check of the fused kernels' inline-asm prefetch: on every control-flow path between an asm
`global_load` and the next asm `s_waitcnt vmcnt`, no instruction may read or overwrite the load's
destination registers (the compiler does not know they are still in flight).  Works on the basic-block
graph of each kernel (labels, s_branch / s_cbranch_*, fall-through) with a forward "registers in
flight" data-flow to a fixed point.  Usage: check_asm_loads.py <kernel.s>"""
import re
import sys

LOAD = re.compile(r"global_load_dword(?:x[234])? ([va])(?:\[(\d+):(\d+)\]|(\d+))")


def regs_of(t):
    regs = set()   # arch VGPR n -> n, accumulation VGPR n -> 1000 + n
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", t):
        base = 1000 if m.group(1) == "a" else 0
        regs.update(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    for m in re.finditer(r"\b([va])(\d+)\b", t):
        regs.add((1000 if m.group(1) == "a" else 0) + int(m.group(2)))
    return regs


def kernels(lines):
    """(name, [(lineno, text, in_asm)]) per function body, up to its s_endpgm"""
    out, cur, name, in_asm = [], None, None, False
    for ln, text in enumerate(lines, 1):
        t = text.strip()
        m = re.match(r"^(_Z\w+):", text)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is None:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith(".Lfunc_end"):
            out.append((name, cur))
            cur = None
            continue
        if not t or t.startswith(";") or (t.startswith(".") and not t.endswith(":")):
            continue
        cur.append((ln, t, in_asm))
    return out


def check(name, body):
    # basic blocks
    label_at, blocks, cur = {}, [], []
    for item in body:
        ln, t, _ = item
        if t.endswith(":"):
            if cur:
                blocks.append(cur)
                cur = []
            label_at[t[:-1]] = len(blocks)
            continue
        cur.append(item)
        if t.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, b in enumerate(blocks):
        last = b[-1][1] if b else ""
        s = []
        m = re.match(r"s_c?branch\w*\s+(\S+)", last)
        if m and m.group(1) in label_at:
            s.append(label_at[m.group(1)])
        if not last.startswith(("s_branch", "s_endpgm", "s_setpc")) and i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    inflight_in = [dict() for _ in blocks]      # reg -> line of the load
    reports = {}
    work = [0] if blocks else []
    seen_in = [None] * len(blocks)
    while work:
        i = work.pop()
        state = dict(inflight_in[i])
        for ln, t, in_asm in blocks[i]:
            if in_asm and t.startswith("s_waitcnt vmcnt"):
                state.clear()
                continue
            if t.startswith("s_endpgm"):
                if state:
                    reports[ln] = f"line {ln}: kernel ends with loads never waited for (loaded at {sorted(set(state.values()))})"
                continue
            m = LOAD.match(t) if in_asm else None
            if m:
                base = 1000 if m.group(1) == "a" else 0
                lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(2) else (int(m.group(4)), int(m.group(4)))
                for r in range(base + lo, base + hi + 1):
                    if r in state:
                        reports[ln] = f"line {ln}: load overwrites in-flight v{r} (loaded at line {state[r]})"
                    state[r] = ln
                continue
            if t.startswith("s_"):
                continue
            hit = regs_of(t) & set(state)
            if hit:
                reports[ln] = f"line {ln}: `{t}` touches in-flight {sorted(hit)} (loaded at {sorted(set(state[r] for r in hit))})"
        for j in succ[i]:
            merged = dict(inflight_in[j])
            merged.update({r: l for r, l in state.items() if r not in merged})
            if seen_in[j] is None or set(merged) != set(inflight_in[j]):
                inflight_in[j] = merged
                seen_in[j] = True
                work.append(j)
    return [reports[k] for k in sorted(reports)]


def check_exec_dpp(body):
    """gfx9 hazard the compiler cannot see inside inline asm: a VALU instruction that writes EXEC (v_cmpx in
    wdpm_stencil.h::select_gt_exec) must be 5 wait states away from the next DPP instruction.  Counted along the
    listing (every instruction one wait state, s_nop N: N + 1) - branches are not followed: the window is 5 instructions."""
    out, since = [], None
    for ln, t, in_asm in body:
        if t.endswith(":"):
            continue
        if "v_cmpx" in t:
            since = 0
            continue
        if since is None:
            continue
        if re.search(r"\b(row_shl|row_shr|row_ror|wave_shl|wave_shr|wave_rol|wave_ror|row_mirror|row_half_mirror|row_bcast|quad_perm|row_newbcast)", t):
            if since < 5:
                out.append(f"line {ln}: DPP `{t}` only {since} wait states after a v_cmpx")
        m = re.match(r"s_nop (\d+)", t)
        since += int(m.group(1)) + 1 if m else 1
        if since >= 5:
            since = None
    return out


def main():
    lines = open(sys.argv[1]).read().splitlines()
    bad = 0
    for name, body in kernels(lines):
        for r in check(name, body) + check_exec_dpp(body):
            print(f"{name[:60]}: {r}")
            bad += 1
    # round 4: no kernel of this file may spill - a hot instantiation that went to scratch would still pass every
    # parity test.  The metadata lists .name / .private_segment_fixed_size per kernel.
    text = "\n".join(lines)
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text):
        if int(m.group(2)):
            print(f"{m.group(1)[:100]}: {m.group(2)} bytes of scratch")
            bad += 1
    # ... and none may outgrow the occupancy its launch is sized for: the relay kernels run two eight-wave workgroups per CU at
    # <= 128 VGPRs (round 4: a second copy of the drain stages took them to 150 - one workgroup per CU, 17 % slower, and every
    # test still passed); the marching kernel two waves per SIMD at <= 256 (its -0.0-safe drain variant is built for one).
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
        name, n = m.group(1), int(m.group(2))
        limit = 128 if "relay_iteration_kernel" in name else (256 if "fused_iteration_kernel" in name and "ILi2ELb1E" not in name else 512)
        if n > limit:
            print(f"{name[:100]}: {n} VGPRs, more than the {limit} its launches are sized for")
            bad += 1
    print("asm prefetch check:", "OK" if not bad else f"{bad} violations")
    sys.exit(1 if bad else 0)


main()
