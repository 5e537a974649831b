#!/usr/bin/env python3
"""Static instruction statistics of a marked region of /tmp/probe/probe.s (built by tools/probe_kernel.sh with -DRSX_ASM_MARKS):
tools/step_stats.py ["mesh steps"] — counts vector / scalar / LDS / SMEM instructions between '; MARK <name> begin' and '; MARK <name> end',
the structurizer's flag branches (s_and(n2)_b64 vcc, exec, ...) and the scalar-spill traffic (v_readlane / v_writelane)."""
import re, sys
name = sys.argv[1] if len(sys.argv) > 1 else "mesh steps"
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/probe/probe.s"
lines = open(path).read().split("\n")
b = [i for i, l in enumerate(lines) if "MARK %s begin" % name in l]
e = [i for i, l in enumerate(lines) if "MARK %s end" % name in l]
for bi, ei in zip(b, e):
    v = s = ds = sm = flag = rl = wl = br = 0
    for l in lines[bi:ei]:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith("v_readlane"): rl += 1
        if op.startswith("v_writelane"): wl += 1
        if op.startswith("v_"): v += 1
        elif op.startswith("ds_"): ds += 1
        elif op.startswith(("s_load", "s_buffer")): sm += 1
        elif op.startswith("s_"):
            s += 1
            if op.startswith("s_cbranch") or op == "s_branch": br += 1
            if re.match(r"s_andn?2?_b64", op) and "vcc, exec" in l: flag += 1
    print("%s: lines %d-%d  V %d  S %d (branches %d, flag-branches %d)  DS %d  SMEM %d  readlane %d writelane %d" % (name, bi, ei, v, s, br, flag, ds, sm, rl, wl))
