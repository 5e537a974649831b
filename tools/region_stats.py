#!/usr/bin/env python3
"""Static instruction counts of ONE kernel in /tmp/probe/probe.s (tools/probe_kernel.sh output) by source file and line range:
tools/region_stats.py <mangled-name-prefix> [file:first-last ...]   — without ranges: per file, and the 40 heaviest lines."""
import collections
import re
import sys

want = sys.argv[1]
cur = None; files = {}; cnt = collections.Counter(); spill = collections.Counter(); inside = False
for line in open('/tmp/probe/probe.s'):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]; continue
    m = re.match(r'^(_Z\w+):', line)
    if m: inside = m.group(1).startswith(want); continue
    if line.startswith('.Lfunc_end'): inside = False
    if not inside: continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m: cur = (files.get(int(m.group(1)), '?'), int(m.group(2))); continue
    m = re.match(r'\s+([vs]_\w+|scratch_\w+|global_\w+|ds_\w+|buffer_\w+|flat_\w+)', line)
    if m and cur:
        cnt[cur] += 1
        if m.group(1).startswith('scratch'): spill[cur] += 1
print("total %d instructions, %d scratch" % (sum(cnt.values()), sum(spill.values())))
if len(sys.argv) > 2:
    for a in sys.argv[2:]:
        f, r = a.split(':'); lo, hi = (int(x) for x in r.split('-'))
        print("%-28s %5d instr %4d scratch" % (a, sum(v for (ff, l), v in cnt.items() if ff == f and lo <= l <= hi), sum(v for (ff, l), v in spill.items() if ff == f and lo <= l <= hi)))
else:
    by = collections.Counter()
    for (f, l), v in cnt.items(): by[f] += v
    print(by.most_common(12))
    for (f, l), v in cnt.most_common(40): print("  %s:%d %d (%d scratch)" % (f, l, v, spill[(f, l)]))
