#!/usr/bin/env python3
"""Register-pressure map of one kernel from hipcc's device assembly (tuning aid).

    hipcc --offload-arch=gfx950 -O3 ... -gline-tables-only --offload-device-only -S rsx_device.hip -o dev.s
    tools/vgpr_live.py dev.s _Z14k_render_traceILb0E [top] [file:first-last,file:first-last,...]

Backward liveness over the kernel's basic blocks on the VGPR operands (first operand of a VALU / load instruction = definition,
everything else = use; compares, stores, readlane define no VGPR), then the live count at every instruction, reported per source
line (max over the instructions attributed to it by the .loc directives)."""
import collections
import re
import sys

path, name = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(name) and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
RX = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for a, b, c in RX.findall(tok):
        if a:
            out.add(int(a))
        else:
            out.update(range(int(b), int(c) + 1))
    return out


NO_DEF = ("v_cmp", "v_cmpx", "global_store", "ds_write", "ds_store", "scratch_store", "buffer_store", "flat_store", "v_readlane",
          "v_readfirstlane", "s_", "global_atomic", "flat_atomic", "ds_bpermute_dummy", "buffer_wbl2", "buffer_inv", "v_nop")
insts = []          # (opcode, defs, uses, loc)
labels = {}
cur = None
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(m.group(1), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"^\s*(\.?L[\w$]+):", l)                  # compiler blocks (.LBBn_m) and the labels of inline assembly (Lname<n>)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    body = l.split(";")[0].strip()
    if not body or body.startswith("."):
        continue
    parts = body.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    if op.startswith(NO_DEF) and not (op.startswith("global_atomic") and "glc" in body):
        d, u = set(), set().union(*[regs(o) for o in ops]) if ops else set()
    else:
        d = regs(ops[0]) if ops else set()
        u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        if op.startswith(("v_writelane", "v_mac", "v_fmac", "v_dot", "v_pk_fmac")) or "op_sel" in body:
            u |= d
    tgt = None
    if op.startswith(("s_cbranch", "s_branch")):
        tgt = ops[0]
    insts.append([op, d, u, cur, tgt])
n = len(insts)
succ = [[] for _ in range(n)]
for i, (op, d, u, loc, tgt) in enumerate(insts):
    if op == "s_endpgm":
        continue
    if op == "s_branch":
        succ[i].append(labels[tgt])
        continue
    if i + 1 < n:
        succ[i].append(i + 1)
    if tgt and tgt in labels:
        succ[i].append(labels[tgt])
live_in = [frozenset()] * n
changed = True
while changed:
    changed = False
    for i in range(n - 1, -1, -1):
        out = set()
        for s_ in succ[i]:
            out |= live_in[s_]
        new = frozenset((out - insts[i][1]) | insts[i][2])
        if new != live_in[i]:
            live_in[i] = new
            changed = True
per_line = collections.defaultdict(int)
count = collections.Counter()
for i in range(n):
    k = insts[i][3]
    per_line[k] = max(per_line[k], len(live_in[i]))
    count[k] += 1
print("instructions %d, peak live VGPRs %d" % (n, max(len(x) for x in live_in)))
for k, v in sorted(per_line.items(), key=lambda kv: -kv[1])[:top]:
    print("%4d live  %-22s line %-5s (%d instructions)" % (v, k[0] if k else "?", k[1] if k else "?", count[k]))

# optional: passengers at the peak — live registers that no instruction of the given source-line ranges touches
if len(sys.argv) > 4:
    ranges = []
    for r in sys.argv[4].split(","):
        name, span = r.split(":")
        first, last = span.split("-")
        ranges.append((name, int(first), int(last)))
    inside = lambda k: k and any(k[0] == f and a <= k[1] <= b for f, a, b in ranges)
    touched = set()
    for op, d, u, loc, tgt in insts:
        if inside(loc):
            touched |= d | u
    peak_i = max(range(n), key=lambda i: len(live_in[i]) if inside(insts[i][3]) else -1)
    live = live_in[peak_i]
    print("peak inside ranges: %d live at %s; %d of them are never touched inside the ranges (passengers)" %
          (len(live), insts[peak_i][3], len(live - touched)))
    # where the passengers were defined / are next used: source lines
    pas = live - touched
    defs = collections.Counter()
    for op, d, u, loc, tgt in insts:
        for r_ in d & pas:
            defs[loc] += 1
    print("passenger definitions by line:", sorted(defs.items(), key=lambda kv: -kv[1])[:25])
    # nearest preceding definition (program order) and nearest following use of every passenger
    where = collections.Counter()
    for r_ in sorted(pas):
        dl = next((insts[i][3] for i in range(peak_i, -1, -1) if r_ in insts[i][1]), None)
        ul = next((insts[i][3] for i in range(peak_i, n) if r_ in insts[i][2] and not inside(insts[i][3])), None)
        where[(dl, ul)] += 1
    for (dl, ul), c in sorted(where.items(), key=lambda kv: -kv[1]):
        print("  %3d regs  defined %-32s next used %s" % (c, dl, ul))
    # context of the global peak: distinct source lines of the 400 instructions around it
    gi = max(range(n), key=lambda i: len(live_in[i]))
    seen = []
    for i in range(max(0, gi - 300), min(n, gi + 150)):
        k = insts[i][3]
        if not seen or seen[-1][0] != k:
            seen.append([k, len(live_in[i]), i])
    print("context of the global peak (instruction %d):" % gi)
    for k, v, i in seen:
        print("   @%5d %4d live  %s" % (i, v, k))
