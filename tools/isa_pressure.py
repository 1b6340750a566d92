#!/usr/bin/env python3
"""VGPR liveness of one kernel in a hipcc -S listing (compile with -gline-tables-only for source lines):

  tools/isa_pressure.py file.s <kernel-substring> [--top N] [--at LINE]

Backward dataflow over the basic blocks of the listing: which VGPRs are live before every instruction.  Prints the peak, the
source lines (from .loc) where the number of live VGPRs is highest, and per source file:line the maximum.  Approximations: every
written register is treated as fully defined (a write under a partial exec mask does not keep the old value alive), branch targets
are the .LBB labels named by s_cbranch_* / s_branch, and indirect control flow does not occur in these kernels.
"""
import re
import sys
from collections import defaultdict

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
NO_DEF_PREFIX = ("global_store", "scratch_store", "buffer_store", "flat_store", "ds_write", "ds_store", "s_", "v_cmp_", "v_cmpx_", "global_atomic", "ds_add_u32 ",
                 "ds_bpermute_never")
# instructions whose first operand is also read (accumulating forms)
READS_DST = ("v_fmac_", "v_mac_", "v_writelane", "v_dot", "v_pk_fmac", "v_cndmask_b32_dpp", "v_mov_b32_dpp", "v_add_u32_dpp", "v_addc", "v_accvgpr")


def regs_of(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse_instr(line):
    t = line.split(";")[0].strip()
    if not t or t.startswith("."):
        return None
    parts = t.split(None, 1)
    op = parts[0]
    ops = parts[1] if len(parts) > 1 else ""
    operands = [o.strip() for o in re.split(r",(?![^\[]*\])", ops)] if ops else []
    defs, uses = set(), set()
    if op.startswith(NO_DEF_PREFIX) and not op.startswith(("s_", "v_cmp")):
        for o in operands:
            uses |= regs_of(o)
        if op.startswith(("global_atomic", "ds_add_rtn", "ds_bpermute", "ds_permute")) and operands:
            pass
    elif op.startswith(("s_", "v_cmp_", "v_cmpx_")):
        for o in operands:
            uses |= regs_of(o)
        if op.startswith("v_readfirstlane") or op.startswith("v_readlane"):
            pass
    elif op.startswith(("v_readfirstlane", "v_readlane")):
        for o in operands[1:]:
            uses |= regs_of(o)
    else:
        if operands:
            defs |= regs_of(operands[0])
            # carry-out forms: v_add_co_u32 v1, vcc, v2, v3 / v_div_scale_f32 v3, s[2:3], ...
            for o in operands[1:]:
                uses |= regs_of(o)
            if op.startswith(READS_DST) or "dpp" in ops or "sdwa" in ops.lower() and False:
                uses |= regs_of(operands[0])
    return op, defs, uses


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
    at = int(sys.argv[sys.argv.index("--at") + 1]) if "--at" in sys.argv else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") is False and ":" in l and not l.startswith("\t"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    blocks, order, cur, loc = {}, [], "entry", ("?", 0)
    blocks[cur] = []
    order.append(cur)
    for l in lines[start + 1:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1); blocks[cur] = []; order.append(cur); continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
        ins = parse_instr(l)
        if ins:
            blocks[cur].append((ins, loc, l.strip()))
    succ = {}
    for i, b in enumerate(order):
        s, fall = set(), True
        for (op, _, _), _, text in blocks[b]:
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = text.split()[1]
                if tgt in blocks:
                    s.add(tgt)
                if op == "s_branch":
                    fall = False
            if op == "s_endpgm":
                fall = False
        if blocks[b] and blocks[b][-1][0][0] == "s_branch":
            fall = False
        elif any(x[0][0] == "s_branch" for x in blocks[b]):
            fall = blocks[b][-1][0][0] != "s_branch"
        if fall and i + 1 < len(order):
            s.add(order[i + 1])
        succ[b] = s
    live_in = {b: set() for b in order}
    changed = True
    while changed:
        changed = False
        for b in reversed(order):
            live = set()
            for s in succ[b]:
                live |= live_in[s]
            for (op, defs, uses), _, _ in reversed(blocks[b]):
                live = (live - defs) | uses
            if live != live_in[b]:
                live_in[b] = live; changed = True
    per_loc = defaultdict(int)
    peak, peak_at = 0, None
    records = []
    for b in order:
        live = set()
        for s in succ[b]:
            live |= live_in[s]
        seq = []
        for ins, loc, text in reversed(blocks[b]):
            op, defs, uses = ins
            after = set(live)
            live = (live - defs) | uses
            n = max(len(live), len(after | defs))
            seq.append((n, loc, text, b, set(live)))
        for rec in reversed(seq):
            records.append(rec)
            n, loc = rec[0], rec[1]
            per_loc[loc] = max(per_loc[loc], n)
            if n > peak:
                peak, peak_at = n, rec
    print("peak live VGPRs: %d at %s:%d  [%s]  %s" % (peak, peak_at[1][0], peak_at[1][1], peak_at[3], peak_at[2]))
    print("source lines with the most live VGPRs:")
    for loc, n in sorted(per_loc.items(), key=lambda kv: -kv[1])[:top]:
        print("  %3d  %s:%d" % (n, loc[0], loc[1]))
    if "--live-at" in sys.argv:
        la = int(sys.argv[sys.argv.index("--live-at") + 1])
        cand = [(i, r) for i, r in enumerate(records) if r[1][1] == la]
        i, rec = max(cand, key=lambda x: x[1][0])
        print("live before `%s` (%s, %d live): register <- source line of the closest earlier write in the listing" % (rec[2], rec[3], rec[0]))
        by_loc = defaultdict(list)
        for r in sorted(rec[4]):
            where = ("?", 0)
            for k in range(i - 1, -1, -1):
                ins = parse_instr(records[k][2])
                if ins and r in ins[1]:
                    where = records[k][1]; break
            by_loc[where].append(r)
        for loc, regs in sorted(by_loc.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
            print("  %s:%d  %s" % (loc[0], loc[1], " ".join("v%d" % r for r in regs)))
    if at is not None:
        print("per instruction at line %d:" % at)
        for n, loc, text, b, live in records:
            if loc[1] == at:
                print("  %3d %-10s %s" % (n, b, text))


if __name__ == "__main__":
    main()
